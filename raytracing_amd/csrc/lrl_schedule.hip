// liblrltrace.so — the C ABI of include/lrl_trace.h: the frame.  The wavefront pipeline (generate, then trace + shade per ray segment until no
// path is alive), its scheduling decisions (hand-over thresholds and their feedback, compaction, sample batches, the timed trial of
// triangle trips) and lrl_render / lrl_render_async.  Host side only; the kernels live in kernels_parity.hip / kernels_fast.hip.
#include "lrl_context.h"

namespace lrl_host {

// ---- wavefront pipeline: generate, then (trace, shade) per ray segment until no path is alive ----
#define LRL_LIST_KINDS 8u        /* ray kinds (closest + one shadow ray per light) the compacted lists are kept for */

// LrlTuning::debug >= 2: a line on stderr before and after every HIP call of the render path, so that a call that blocks names itself
#define CRUMB_TRY(c, what, expr)                                                        \
    do {                                                                                \
        if ((c)->tuning.debug >= 2) { std::fprintf(stderr, "lrl[%d]: > %s\n", (c)->device, what); std::fflush(stderr); } \
        HIP_TRY(c, expr);                                                               \
        if ((c)->tuning.debug >= 2) { std::fprintf(stderr, "lrl[%d]: < %s\n", (c)->device, what); std::fflush(stderr); } \
    } while (0)

// where the blocks of one pass (bsz samples of Ppix path slots each) sit in the arena; every size grows with bsz, so the layout of a
// smaller last batch fits the arena of a full one
struct WfLayout {
    size_t state, flags, shadow, rayQ, rayD, shadowQ, maskC, maskS, maskL, refr, ovf, alive, contHead, contRefs, rayList, listCount, total;
};
static WfLayout wfLayout(size_t P, uint32_t L, uint32_t refrEntries, uint32_t ovfN, uint32_t contSlots, uint32_t contStride, bool special) {
    WfLayout o{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
    const size_t words = P / 64, chunks = words * (1 + size_t(L)) / LRL_WF_CHUNK_WORDS;
    o.state = take(P * 16 * 6); o.flags = take(P * 4); o.shadow = take(P * size_t(L) * 4); o.rayQ = take(P * 16); o.rayD = take(P * 8); o.shadowQ = take(P * size_t(L) * 16);
    o.maskC = take(words * 8); o.maskS = take(words * 8); o.maskL = take(words * 8);
    o.refr = take(P * 64 * refrEntries); o.ovf = take(chunks * ovfN * 512);
    o.alive = take(LRL_WF_MAX_ITERATIONS * 4 * 2 + LRL_FEEDBACK_ITERS * LRL_WORK_COUNTERS * 128);
    o.contHead = take(size_t(contSlots) * 16); o.contRefs = take(size_t(contSlots) * contStride * 4);
    o.rayList = take(special ? P * (1 + size_t(L)) * 4 : 0);
    o.listCount = take(special ? size_t(LRL_WF_MAX_ITERATIONS) * LRL_LIST_KINDS * 4 : 0);
    o.total = off;
    return o;
}

static int renderWavefront(LrlContext* c, const LrlRenderParams* p, const KScene& k, KFrame f, float* rgbaDev, uint32_t* hitDev,
                           unsigned long long* ctr, hipStream_t stream, uint32_t rows, hipEvent_t e0, hipEvent_t e1) {
    const LrlTuning& tn = c->tuning;
    const bool fast = p->mode == LRL_MODE_FAST, pool = p->variant == LRL_VARIANT_POOL;
    const uint32_t L = c->num.pointLightNum;
    const uint32_t tilesX = (c->width + 7) / 8, tilesY = (rows + 7) / 8;
    const uint64_t Ppix64 = (uint64_t(tilesX) * tilesY * 64 + LRL_WF_SLOT_ALIGN - 1) / LRL_WF_SLOT_ALIGN * LRL_WF_SLOT_ALIGN;
    if (Ppix64 * (1 + uint64_t(L)) >= (1ull << 31)) return fail(c, LRL_ERR_LIMIT, "pixels x (1 + lights) of one render exceeds 2^31 ray slots: render in row stripes");
    const size_t Ppix = size_t(Ppix64);
    // samples traced together: one generate / trace / shade pass carries `batch` samples (slot = b * Ppix + pixel slot), so that
    // the launches of a high-spp frame fill the chip (a 1080p sample is 2 M paths; the chip wants about 8 M rays per launch) and a frame
    // pays the tails of its launches once per batch instead of once per sample
    const uint32_t nSamples = f.sampleEnd - f.sampleBegin;
    // (measured, 1080p: C3 132.9 / 131.4 / 131.2 / 128.2 ms per 64-spp frame with batches of 4 / 8 / 16 / 32 samples, 148.8 with one; C5
    // 67.6 / 63.1 / 60.3 ms per 16-spp frame with 4 / 8 / 16).  Default: 2^26 path slots per launch, as long as the arena stays within 24 GiB
    // of the 288 (round 3, with the samples of a pass side by side: C3 64.3 / 63.8 / 63.3 / 63.0 ms with 6 / 8 / 16 / 64 samples a pass, C4
    // 742 / 732 / 723 ms with 4 / 8 / 16; LrlTuning.batchSlots caps it for a caller who needs the memory), the passes of a frame equally long
    const uint64_t batchSlots = tn.batchSlots ? tn.batchSlots : (1u << 26);
    uint32_t batch = uint32_t(std::min<uint64_t>(nSamples, std::max<uint64_t>(1, batchSlots / Ppix64)));
    while (batch > 1 && Ppix64 * batch * (1 + uint64_t(L)) >= (1ull << 31)) --batch;
    if (!tn.batchSlots) {
        const uint64_t perSlot = 16 * 6 + 4 + (1 + uint64_t(L)) * 36 + 64ull * (c->hasTransmission ? p->maxDepth : 0u);
        while (batch > 1 && Ppix64 * batch * perSlot > (24ull << 30)) --batch;
    }
    if (batch > 1) batch = (nSamples + (nSamples + batch - 1) / batch - 1) / ((nSamples + batch - 1) / batch);
    // Trace kernel per iteration.  PARITY and counting renders: wfTraceKernel (the reference's visit order).  FAST: the lean
    // one-lane-per-ray kernel over the four-wide tree, which hands rays that are still going after handoverTrips trips to the
    // lanes-per-ray kernel.
    const bool special = fast && !tn.traceRef && (!ctr || (p->flags & LRL_FLAG_COUNT_FETCHES));
    // deepest stack: one parked child per level of the binary tree (wfTraceKernel), three per level of the four-wide tree (lean)
    const uint32_t stackNeed = special ? std::max(c->bvhDepth + 1, 3 * c->node4Depth + 4) : c->bvhDepth + 1;
    const uint32_t ldsRows = special ? uint32_t(LRL_LEAN_STACK) : uint32_t(LRL_WF_STACK);
    const uint32_t ovfN = stackNeed > ldsRows ? stackNeed - ldsRows : 0u;
    const uint32_t refrEntries = c->hasTransmission ? p->maxDepth : 0u;
    const uint32_t contStride = stackNeed + 1;
    const uint32_t contSlots = tn.contSlots ? std::max(1024u, tn.contSlots)
                                            : uint32_t(std::min<uint64_t>(1u << 22, 262144ull * ((Ppix64 * batch + (1u << 21) - 1) >> 21)));
    const WfLayout full = wfLayout(Ppix * batch, L, refrEntries, ovfN, contSlots, contStride, special);
    int rc = ensure(c, c->wfArena, full.total);
    if (rc) return rc;
    char* const base = static_cast<char*>(c->wfArena.p);
    // base thresholds of the hand-over: a scene in the caches is best off with early hand-overs (C2: a plateau around 48, 48, 32, 16, 16);
    // beyond them a trip of the lean kernel and a round of the hand-over kernel both cost an HBM miss, and the lean kernel's ordered walk
    // needs fewer of them per ray: later (C5 with 16 lanes per ray: 40.7 ms with 48 / 16, 35.3 with 64 / 32, 34.5 with 96 / 48, 35.3 with 128 / 64)
    // A launch of many waves per slot (batches of samples, 4K frames) is bound less by its chains than a 1080p sample: in between (C4, 2^25
    // path slots a launch: 865 ms with 48 / 16, 797 with 64 / 32, 807 with 96 / 48, 844 without hand-over).
    const bool beyondCachesThr = size_t(c->recNum) * 64 > (size_t(1) << 30), bigLaunch = Ppix64 * batch >= (1ull << 23);
    const uint32_t regime = beyondCachesThr ? 2u : bigLaunch ? 1u : 0u;
    if (c->handoverRegime != regime) { c->handoverRegime = regime; c->handoverThr.clear(); c->feedbackTick = 0; c->contIters = 0; }
    const uint32_t handoverEarly = tn.handoverEarly != LRL_HANDOVER_AUTO ? tn.handoverEarly : regime == 2 ? 96u : regime == 1 ? 64u : 48u;
    const uint32_t handoverLate = tn.handoverLate != LRL_HANDOVER_AUTO ? tn.handoverLate : regime == 2 ? 48u : regime == 1 ? 32u : 16u;
    // The hand-over pays for rays with much work left.  Where most rays are of middling length (Pool: flat tree of tall
    // boxes) the slots run out, or the rays that leave are nearly done and cost more to move than to finish, so the
    // threshold of an iteration follows what that iteration did in an earlier pass: doubled (x4 when far beyond
    // the slots) while (nearly) all slots were asked for or the handed-over rays needed fewer than workLow frontier
    // entries each.  A ratchet: thresholds that came back down made the Pool oscillate between two states from sample to
    // sample (160 ms a frame against 150 for any pinned state); a new scene starts from the defaults.  adapt = 0 pins them.
    // lanes per handed-over ray: groups of 16 when a group's share of the frontier holds the hand-over plus a depth-first
    // descent (lrl_lean.inc), else a wave per ray
    // Round 2 measured 49.1 ms on C5 with a wave per ray against 60.0 with 16 lanes, and made the wave the default beyond the caches; with the
    // later hand-over above and 512-entry frontiers (capX below) it is the other way round, because what bounds the rounds there is their latency
    // and four rays a wave are four times the rays in flight: C5 44.3 ms with a wave per ray, 34.5 with 16 lanes (32 lanes: 35.0; 8: 34.3).
    const bool beyondCaches = size_t(c->recNum) * 64 > (size_t(1) << 30);
    uint32_t swarmGroup = tn.groupLanes ? tn.groupLanes : 16u;
    // Four lanes to a record fetch (lrl_lean.inc).  A uniformly random gather out of 16 GiB runs 2.3x faster that way (address translations
    // per lane bound it: tools/probes/hbm_gather.hip), but the rays of C5 are not that incoherent: 54.2 against 48.5 ms per frame (every kernel
    // of the frame ~15 % slower: idle lanes fetch too, 8 shuffles + 16 swaps a trip).  Off unless asked for.
    const bool coop = tn.coopFetch == 1;
    const bool groupQuad = tn.groupQuad == 1;      // lanes-per-ray kernel: four lanes to a frontier entry (the wave-per-ray kernel switches per round by itself)
    // Triangle trips (lrl_lean.inc, PH): -7 % on the Pool and -3 % on C5, +3-5 % on the coherent mirror rays of C2 / C4, and nothing the host knows
    // beforehand tells the two apart.  The frames are the same bit for bit either way, so the library times frames each way (events on the
    // caller's stream, read back without waiting) and keeps the faster for this scene and frame shape (two frames each way, alternating, after three that let the thresholds settle).  phasedTrips = 1 / 2 pins it.
    LrlContext::PhasedTrial& trial = c->phasedTrial;
    hipEvent_t trialEnd = nullptr;
    bool phased = false;
    uint32_t trialState = 0;
    const bool deterministic = tn.deterministic != 0;
    if (special && !coop) {
        if (tn.phasedTrips) phased = tn.phasedTrips == 1;
        else if (deterministic) phased = false;
        else {
            const uint64_t signature = (Ppix64 * batch) ^ (uint64_t(f.sampleEnd - f.sampleBegin) << 32) ^ (uint64_t(p->maxDepth) << 48) ^ (uint64_t(p->flags & LRL_FLAG_DIFFUSE_BOUNCE) << 52) ^
                                       (uint64_t(pool) << 57) ^ (uint64_t(L) << 58) ^ (ctr ? 1ull << 63 : 0ull);
            if (trial.signature != signature) { trial.signature = signature; trial.state = 0; }
            if (trial.state == 7 && hipEventQuery(trial.ev[7]) == hipSuccess) {
                float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                bool ok = true;
                for (int i = 0; i < 4; ++i) ok = ok && hipEventElapsedTime(&t[i], trial.ev[2 * i], trial.ev[2 * i + 1]) == hipSuccess;
                trial.choice = ok && t[1] + t[3] < 0.99f * (t[0] + t[2]);
                trial.state = 8;
                if (tn.debug == 1 || tn.debug >= 3) std::fprintf(stderr, "lrl: triangle trips %s (frames %.3f, %.3f ms with, %.3f, %.3f ms without)\n", trial.choice ? "on" : "off", t[1], t[3], t[0], t[2]);
            }
            if (trial.state >= 3 && trial.state <= 6) {
                const int k = trial.state - 3;
                phased = (k & 1) != 0;
                CRUMB_TRY(c, "hipEventRecord(trial start)", hipEventRecord(trial.ev[2 * k], stream));
                trialEnd = trial.ev[2 * k + 1];
            } else if (trial.state == 8) phased = trial.choice;
            if (trial.state < 7) ++trial.state;
            trialState = trial.state == 8 ? 2u : 1u;
        }
    }
    // lane refill (wfTraceLeanStreamKernel): waves that own streamWords words of a launch
    const uint32_t streamWords = tn.streamWords, refillMin = tn.refillMin ? std::min(64u, tn.refillMin) : 16u;
    if (swarmGroup != 8 && swarmGroup != 16 && swarmGroup != 32) swarmGroup = 64;
    // a ray's frontier: LRL_GROUP_CAP entries, twice that where the hand-over record + depth-first reserve (contStride) would
    // leave less than 2/5 of it for full rounds (groups of 16 lanes only)
    const uint32_t capX = tn.groupCapX ? tn.groupCapX : (swarmGroup == 16 && 2 * contStride + 3 * swarmGroup > LRL_GROUP_CAP - 32) ? 2u : 1u;      // C2 (stride 77): 256 is 1.3 % faster; 100 M device-built tree (95): 512 is 1.85x faster on C5
    const bool instantiated = (swarmGroup == 16 && (capX == 1 || capX == 2 || capX == 4)) || (swarmGroup == 32 && (capX == 1 || capX == 2 || capX == 4)) || (swarmGroup == 8 && capX == 1);
    const uint32_t groupCap = LRL_GROUP_CAP * (instantiated ? capX : 1u);
    if (swarmGroup != 64 && (!instantiated || 2 * contStride + 3 * swarmGroup > groupCap)) swarmGroup = 64;
    const bool handoverFits = 2 * contStride + 192 <= LRL_SWARM_CAP;      // else (trees hundreds of levels deep) the lean kernel finishes every ray itself
    const uint32_t sparseFrom = tn.sparseFrom;      // (measured: C2 best with none of its 5, C1 with 8 of its 56, C3 with 4 of its 10)
    const bool adapt = tn.adapt != 0 && !deterministic;
    // few rays left after the first bounces: the waves are free sooner (C2: a plateau around 48,48,32,16,16).  Beyond the caches the
    // descent starts one iteration earlier (96, 64, 48, ...): C5 32.9 -> 32.2 ms, the 100 M-triangle scene with mirror bounces 2.54 -> 2.44
    // (96, 72, 48: 2.45; 96, 96, 72, 48 as in the caches: the former default; C2 with diffuse bounces loses with anything below 48, 48, 32)
    auto baseThr = [&](uint32_t i) {
        if (tn.handoverListLen) return tn.handoverList[std::min<uint32_t>(i, tn.handoverListLen - 1)];
        if (regime == 2) return i == 0 ? handoverEarly : i == 1 ? (handoverEarly + 2 * handoverLate) / 3 : handoverLate;
        return i < 2 ? handoverEarly : i == 2 ? (handoverEarly + handoverLate) / 2 : handoverLate;
    };
    if (c->handoverThr.empty()) {
        c->handoverThr.resize(LRL_FEEDBACK_ITERS);
        for (uint32_t i = 0; i < LRL_FEEDBACK_ITERS; ++i) c->handoverThr[i] = baseThr(i);
        // (until a feedback pass says how many rays an iteration hands over, and for good in deterministic mode: the kernel strides over the
        // slots whatever the grid, and dispatching one workgroup per slot costs 82 us a launch on the Pool's 1.5 M slots, most of them idle)
        c->swarmGrid.assign(LRL_FEEDBACK_ITERS, std::min(contSlots, 1u << 17));
        c->handoverIdle.assign(LRL_FEEDBACK_ITERS, 0);
    }
    const float workLow = tn.workLow;
    c->workPerRay.resize(LRL_FEEDBACK_ITERS, 0.0f);
    // The counters of a pass reach the host without a copy: the shade kernel's first block stores them into pinned memory
    // (WFrame::feedbackHost) and an event recorded behind the pass says when they are all there.  (Round 1 copied them with
    // hipMemcpyAsync on the caller's stream; a device-to-host copy is the one call of the render path that may wait for the
    // stream to drain inside the runtime, and a caller's stream may itself be waiting for a collective.)
    auto feedback = [&]() {
        if (!c->contIters || hipEventQuery(c->evCont) != hipSuccess) return;
        const uint32_t slotsThen = c->contSlotsInFlight;
        for (uint32_t i = 0; i < c->contIters; ++i) {
            uint32_t& t = c->handoverThr[i];
            const volatile uint32_t* h = c->contHost + size_t(i) * 8;
            const uint32_t asked = h[0];
            c->handoverIdle[i] = asked == 0u;
            if (t == 0) continue;
            uint64_t work = 0;
            for (uint32_t w = 0; w < LRL_WORK_COUNTERS; ++w) work += h[1 + w];
            const uint32_t served = std::min(asked, slotsThen);
            const float perRay = served ? 16.0f * float(work) / float(served) : 0.0f;       // one wave in 16 reports
            c->workPerRay[i] = perRay;
            // a few dozen rays that were nearly done (the default scene: 20-90 rays an iteration with 2-5 entries each) are not worth a launch
            if (served < 1024u && perRay < workLow) c->handoverIdle[i] = 1;
            // rays that were nearly done when they left (a few frontier entries each) cost more to hand over than to finish
            // where they were: Pool's rays are all of middling length; the mesh scenes' long rays need 30-40 entries
            if (!adapt) {}
            else if (served >= 1024 && perRay < workLow) t = std::min(t * 2, 4096u);
            else if (asked > 4 * uint64_t(slotsThen)) t = std::min(t * 4, 4096u);
            else if (asked > slotsThen / 16 * 15) t = std::min(t * 2, 4096u);
            // dispatching 65536 workgroups that find nothing costs ~15 us a launch: twice what was asked for last time is
            // plenty (the kernel strides over the slots whatever the grid)
            c->swarmGrid[i] = std::max(2048u, uint32_t(std::min<uint64_t>(1u << 22, (2ull * asked + 1023u) / 1024u * 1024u)));
        }
        c->contIters = 0;
    };
    // compaction of the live rays before the trace launches of the late segments (compactFrom: first such iteration,
    // 0 = never).  It pays where the survivors are scattered over the mask words or a trip is expensive: after diffuse
    // bounces (C2 with them: 2.07 -> 1.60 ms) and on scenes far beyond the caches (100 M triangles: 2.84 -> 2.62 ms; C5:
    // 141.5 -> 132.6 ms); on C2 itself, the Pool and the default scene the extra kernel per iteration costs 1.5-10 %.
    // With more kinds of rays than LRL_LIST_KINDS (lights + 1) the masks are used throughout.
    const uint32_t compactFrom = (1 + L > LRL_LIST_KINDS) ? 0u : tn.compactFrom >= 0 ? uint32_t(tn.compactFrom)
                                 : (f.diffuse || beyondCaches) ? 2u : 0u;
#ifdef LRL_DIAG
    const char* tlPath = std::getenv("LRL_TIMELINE");          // diagnostic builds only: timeline of ONE trace launch
    const char* tlIter = std::getenv("LRL_TIMELINE_ITER");
    const uint32_t tlWanted = tlIter ? uint32_t(std::atoi(tlIter)) : 0u;
    const size_t tlEntries = std::max<size_t>(Ppix * batch * (1 + size_t(L)) / 16, 131072u + 65536u);      // lean: one per mask word; wave-per-ray: 131072 + slot
    if (tlPath && *tlPath && (rc = ensure(c, c->timelineDev, tlEntries * 48))) return rc;
#endif
    // every path ends after at most 2^(maxDepth+1) segments (each hit continues one reflection and parks one refraction)
    // (without a transmitting material nothing is ever parked: a path is primary + at most maxDepth reflections, and its
    // last segment casts no shadow rays, so iteration maxDepth finalises whatever is left)
    const uint32_t bound = c->hasTransmission ? uint32_t(std::min<uint64_t>(LRL_WF_MAX_ITERATIONS, (2ull << p->maxDepth) + 2)) : p->maxDepth + 1;
    const bool perKernel = c->timing && (p->flags & LRL_FLAG_TIME_KERNELS);
    const uint32_t pollLag = std::min(3u, std::max(1u, tn.pollLag));
    const bool poll = c->hasTransmission;    // refraction stacks make the number of segments data dependent
    // the rest of a pass in one launch (lrl_lean.inc, wfTailKernel): by default where a material transmits, from the sixth segment on
    const uint32_t tailFrom = tn.tailFrom ? tn.tailFrom : (poll ? 5u : LRL_TAIL_NEVER);
    const bool useTail = special && tailFrom != LRL_TAIL_NEVER && !ctr;
    // ... but hardly from one pass (or frame) to the next: as many iterations as the last pass needed are enqueued without looking, the
    // host looks once at the end of that stretch, and goes on one iteration at a time only if paths are still alive.  (Looking every
    // iteration kept the host one iteration ahead at most, and the late iterations of a pass are shorter than a wake-up of the host: the
    // Pool at 64 samples spent 13 of its 70 ms per frame between kernels.)  tuning.pollLag = 0 looks every iteration, as before.
    uint32_t& expected = c->expectedIters;
    if (e0) CRUMB_TRY(c, "hipEventRecord(frame start)", hipEventRecord(e0, stream));
    uint32_t iterations = 0, passIndex = 0;
    for (uint32_t s = f.sampleBegin; s < f.sampleEnd; ++passIndex) {
        const uint32_t bsz = std::min(batch, f.sampleEnd - s);
        const size_t P = Ppix * bsz;
        const uint32_t words = uint32_t(P / 64);
        const WfLayout lay = wfLayout(P, L, refrEntries, ovfN, contSlots, contStride, special);
        WFrame wf{};
        wf.P = uint32_t(P); wf.words = words; wf.tilesX = tilesX; wf.tilesY = tilesY; wf.overflowEntries = ovfN;
        wf.sample = s; wf.batch = bsz; wf.Ppix = uint32_t(Ppix); wf.blocksPerSample = uint32_t(Ppix / 256);
        float4* st = reinterpret_cast<float4*>(base + lay.state);
        wf.answer = st; wf.ratio = st + P; wf.decay = st + 2 * P;
        wf.hit = reinterpret_cast<float2*>(st + 3 * P); wf.hitAux = reinterpret_cast<float*>(wf.hit + P);      // (8 + 4 of the 16 B a path)
        wf.shadow = reinterpret_cast<float*>(base + lay.shadow);
        wf.pendA = st + 4 * P; wf.pendB = st + 5 * P;
        wf.flags = reinterpret_cast<uint32_t*>(base + lay.flags);
        wf.rayQ = reinterpret_cast<float4*>(base + lay.rayQ);
        wf.rayD = reinterpret_cast<float2*>(base + lay.rayD);
        wf.shadowQ = reinterpret_cast<float4*>(base + lay.shadowQ);
        wf.maskC = reinterpret_cast<unsigned long long*>(base + lay.maskC);
        wf.maskS = reinterpret_cast<unsigned long long*>(base + lay.maskS);
        wf.maskL = reinterpret_cast<unsigned long long*>(base + lay.maskL);
        wf.refr = refrEntries ? reinterpret_cast<float4*>(base + lay.refr) : nullptr;
        wf.overflow = reinterpret_cast<uint2*>(base + lay.ovf);
        wf.alive = reinterpret_cast<uint32_t*>(base + lay.alive);
        uint32_t* const contCounts = wf.alive + LRL_WF_MAX_ITERATIONS;      // one per iteration, zeroed with `alive`
        uint32_t* const contWorkAll = contCounts + LRL_WF_MAX_ITERATIONS;   // [iteration < LRL_FEEDBACK_ITERS][LRL_WORK_COUNTERS] 128 B apart, zeroed with `alive`
        wf.handoverTrips = handoverEarly;
        wf.contSlots = contSlots; wf.contStride = contStride; wf.groupCap = groupCap;
        wf.contHead = reinterpret_cast<uint4*>(base + lay.contHead);
        wf.contRefs = reinterpret_cast<uint32_t*>(base + lay.contRefs);
        wf.contCount = contCounts;
        wf.contWork = contWorkAll;
        wf.rayList = reinterpret_cast<uint32_t*>(base + lay.rayList);
        uint32_t* const listCounts = reinterpret_cast<uint32_t*>(base + lay.listCount);      // [iteration][LRL_LIST_KINDS]
        wf.listCount = listCounts;
        wf.out = reinterpret_cast<float4*>(rgbaDev);
        wf.hitOut = reinterpret_cast<uint2*>(hitDev);
        wf.counters = ctr;
        wf.timeline = nullptr;
        if (special && !deterministic) feedback();
        // the feedback need not be fresh: the first passes after a scene upload, then one pass in eight (and never two in flight)
        const bool report = special && !deterministic && c->contIters == 0 && (c->feedbackTick++ < 8 || (c->feedbackTick & 7u) == 0);
        wf.feedbackHost = report ? c->contHostDev : nullptr;
        if (report) std::memset(c->contHost, 0, size_t(LRL_FEEDBACK_ITERS) * 32);
        // alive, hand-over counts, work counters and the list counts of compacted iterations: cleared by the generate kernel
        wf.zeroWords = special ? LRL_WF_MAX_ITERATIONS * 2u + LRL_FEEDBACK_ITERS * LRL_WORK_COUNTERS * 32u : LRL_WF_MAX_ITERATIONS;
        wf.zeroListWords = (special && compactFrom != 0 && bound > compactFrom) ? bound * LRL_LIST_KINDS : 0u;
        // polled words: the half of the pinned array that pass - 2 used (the host has seen that pass end)
        uint32_t* const aliveWords = c->aliveHost + size_t(passIndex & 1u) * LRL_WF_MAX_ITERATIONS;
        wf.aliveHost = poll ? c->aliveHostDev + size_t(passIndex & 1u) * LRL_WF_MAX_ITERATIONS : nullptr;
        if (poll) std::memset(aliveWords, 0, size_t(bound) * 4);
        CRUMB_TRY(c, "generate", fast ? lrl_fast::wfLaunchGenerate(k, f, wf, stream) : lrl_parity::wfLaunchGenerate(k, f, wf, stream));
        uint32_t iter = 0;
        const uint32_t stretch = (poll && tn.pollLag != 0) ? std::min(expected, bound) : 0u;
        bool tailDone = false;
        uint32_t tailStart = bound;
        for (; iter < bound; ++iter) {
#ifdef LRL_DIAG
            const bool tl = tlPath && *tlPath && iter == tlWanted && s == f.sampleBegin;
            if (tl) { HIP_TRY(c, hipMemsetAsync(c->timelineDev.p, 0, tlEntries * 48, stream)); wf.timeline = static_cast<unsigned long long*>(c->timelineDev.p); }
#endif
            hipEvent_t ta = nullptr, tb = nullptr;
            if (perKernel && c->timingUsed < LRL_MAX_TIMED_LAUNCHES) {
                if (c->timingUsed == c->timingEvents.size()) {
                    HIP_TRY(c, hipEventCreate(&ta));
                    HIP_TRY(c, hipEventCreate(&tb));
                    c->timingEvents.emplace_back(ta, tb);
                }
                ta = c->timingEvents[c->timingUsed].first; tb = c->timingEvents[c->timingUsed].second;
                ++c->timingUsed;
                HIP_TRY(c, hipEventRecord(ta, stream));
            }
            if (useTail && iter >= tailFrom) {
                tailStart = iter;
                // the rest of the pass in one launch (wfTailKernel): every wave finishes its own paths; then one look at the words it left
                CRUMB_TRY(c, "tail", lrl_fast::wfLaunchTail(k, f, wf, iter, bound, pool, stream));
                if (tb) HIP_TRY(c, hipEventRecord(tb, stream));      // (timed like a trace launch; it shades too)
                tailDone = true;
                if (!poll) { iter = bound; break; }      // an exact bound: nothing to ask the device
                CRUMB_TRY(c, "hipEventRecord(tail)", hipEventRecord(c->evIter[0], stream));
                CRUMB_TRY(c, "hipEventSynchronize(tail)", hipEventSynchronize(c->evIter[0]));
                uint32_t last = iter;
                while (last < bound && reinterpret_cast<volatile uint32_t*>(aliveWords)[last] != 0) ++last;
                if (last == bound) return fail(c, LRL_ERR_LIMIT, "wavefront pipeline did not finish within its iteration bound");
                iter = last + 1;
                break;
            }
            if (special) {
                const uint32_t fi = std::min(iter, LRL_FEEDBACK_ITERS - 1);
                wf.contCount = contCounts + (iter < LRL_WF_MAX_ITERATIONS ? iter : 0);
                // an iteration whose rays all finished below the threshold at the last feedback runs without hand-over (the same rays, finished
                // by the lean kernel as they were anyway) and without the launch that would find nothing: the default scene has 56 such
                // iterations a frame, 9 us each; every feedback pass looks again
                wf.handoverTrips = (handoverFits && (report || !c->handoverIdle[fi])) ? c->handoverThr[fi] : 0u;
                wf.contWork = contWorkAll + size_t(fi) * LRL_WORK_COUNTERS * 32;
                const bool compact = compactFrom != 0 && iter >= compactFrom && iter < LRL_WF_MAX_ITERATIONS;
                wf.listCount = listCounts + size_t(iter < LRL_WF_MAX_ITERATIONS ? iter : 0) * LRL_LIST_KINDS;
                CRUMB_TRY(c, "trace (lean)", lrl_fast::wfLaunchTraceLean(k, wf, iter >= sparseFrom, compact, coop, phased, streamWords, refillMin, stream));
                if (wf.handoverTrips) CRUMB_TRY(c, "trace (hand-over)", lrl_fast::wfLaunchTraceSwarm(k, wf, std::min(contSlots, c->swarmGrid[fi]), swarmGroup, coop, groupQuad, stream));
            }
            else CRUMB_TRY(c, "trace", fast ? lrl_fast::wfLaunchTrace(k, wf, stream) : lrl_parity::wfLaunchTrace(k, wf, stream));
            if (tb) HIP_TRY(c, hipEventRecord(tb, stream));
#ifdef LRL_DIAG
            if (tl) {
                wf.timeline = nullptr;
                std::vector<unsigned long long> h(tlEntries * 6);
                HIP_TRY(c, hipStreamSynchronize(stream));
                HIP_TRY(c, hipMemcpy(h.data(), c->timelineDev.p, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* fp = std::fopen(tlPath, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
            }
#endif
            CRUMB_TRY(c, "shade", fast ? lrl_fast::wfLaunchShade(k, f, wf, iter, pool, stream) : lrl_parity::wfLaunchShade(k, f, wf, iter, pool, stream));
            if (!poll) continue;   // exact bound: nothing to ask the device, the whole frame is enqueued without a host wait
            if (iter + 1 < stretch) continue;
            CRUMB_TRY(c, "hipEventRecord(iteration)", hipEventRecord(c->evIter[iter & 3], stream));
            if (iter + 1 == stretch) {
                CRUMB_TRY(c, "hipEventSynchronize(stretch)", hipEventSynchronize(c->evIter[iter & 3]));
                if (reinterpret_cast<volatile uint32_t*>(aliveWords)[iter] == 0) { ++iter; break; }
                continue;
            }
            if (iter >= pollLag) {      // look at an earlier iteration while this one runs: the pipeline ends pollLag (empty) iterations late
                CRUMB_TRY(c, "hipEventSynchronize(iteration)", hipEventSynchronize(c->evIter[(iter - pollLag) & 3]));
                if (reinterpret_cast<volatile uint32_t*>(aliveWords)[iter - pollLag] == 0) { ++iter; break; }
            }
        }
        if (poll && iter == bound && !tailDone) {      // the last iteration has not been looked at yet
            CRUMB_TRY(c, "hipEventSynchronize(last iteration)", hipEventSynchronize(c->evIter[(bound - 1) & 3]));
            if (reinterpret_cast<volatile uint32_t*>(aliveWords)[bound - 1] != 0) return fail(c, LRL_ERR_LIMIT, "wavefront pipeline did not finish within its iteration bound");
        }
        if (poll) {      // what this pass needed: up to the first iteration that left nothing alive (every word up to it has been seen written)
            uint32_t need = 0;
            while (need < iter && reinterpret_cast<volatile uint32_t*>(aliveWords)[need] != 0) ++need;
            expected = std::min(need + 1, iter);
        }
        if (bsz > 1) CRUMB_TRY(c, "accumulate", fast ? lrl_fast::wfLaunchAccumulate(f, wf, stream) : lrl_parity::wfLaunchAccumulate(f, wf, stream));
        iterations += iter;
        if (report) {
            c->contIters = std::min(std::min(iter, tailStart), LRL_FEEDBACK_ITERS);      // (the tail launch hands nothing over and reports nothing)
            c->contSlotsInFlight = contSlots;
            CRUMB_TRY(c, "hipEventRecord(feedback)", hipEventRecord(c->evCont, stream));
        }
        if ((tn.debug == 1 || tn.debug >= 3) && special) {      // (2: only the lines around the HIP calls, no added waits)
            std::vector<uint32_t> cc(iter);
            HIP_TRY(c, hipStreamSynchronize(stream));
            HIP_TRY(c, hipMemcpy(cc.data(), contCounts, iter * 4, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "lrl: samples %u..%u, handed-over rays per iteration (threshold, entries per ray at the last feedback):", s, s + bsz);
            for (uint32_t i = 0; i < iter; ++i) std::fprintf(stderr, " %u (%u, %.0f)", cc[i], c->handoverThr[std::min(i, LRL_FEEDBACK_ITERS - 1)], c->workPerRay[std::min(i, LRL_FEEDBACK_ITERS - 1)]);
            std::fprintf(stderr, "  [%u lanes per handed-over ray, stride %u, %u slots]\n", swarmGroup, contStride, contSlots);
        }
        s += bsz;
    }
    if (trialEnd) CRUMB_TRY(c, "hipEventRecord(trial end)", hipEventRecord(trialEnd, stream));
    if (e1) CRUMB_TRY(c, "hipEventRecord(frame end)", hipEventRecord(e1, stream));
    c->lastIterations = iterations;
    LrlRenderInfo& ri = c->renderInfo;
    ri = LrlRenderInfo{};
    ri.iterations = iterations; ri.passes = passIndex; ri.batch = batch;
    if (special) {
        ri.compactFrom = compactFrom; ri.coopFetch = coop ? 1u : 0u; ri.streamWords = coop ? 0u : streamWords; ri.phasedTrips = phased ? 1u : 0u; ri.groupLanes = swarmGroup; ri.frontierEntries = swarmGroup == 64 ? LRL_SWARM_CAP : groupCap;
        ri.contSlots = contSlots; ri.contStride = contStride;
        for (uint32_t i = 0; i < 8; ++i) ri.handoverTrips[i] = handoverFits ? c->handoverThr[i] : 0u;
        ri.trialState = trialState;
    }
    return LRL_OK;
}

static int renderImpl(LrlContext* c, const LrlRenderParams* p, float* rgbaDev, uint32_t* hitDev, hipStream_t stream, uint32_t rows) {
    const uint32_t spp = p->spp;
    uint32_t s0 = p->sampleBegin, s1 = p->sampleEnd;
    if (s0 == 0 && s1 == 0) s1 = spp;
    // per-sample camera: the jitter is the shear of `trans`; computed here in float like the host helper
    std::vector<LrlTrans> tr(spp, c->trans);
    if (p->jitter) {
        for (uint32_t s = 0; s < spp; ++s) {
            float dx = p->jitter[s * 2], dy = p->jitter[s * 2 + 1];
            if (dx == 0.0f && dy == 0.0f) continue;
            for (int i = 0; i < 3; ++i)
                tr[s].trans[i][2] = c->trans.trans[i][2] + (c->trans.trans[i][0] * (2.0f * dx / c->trans.z0) + c->trans.trans[i][1] * (2.0f * dy / c->trans.z0));
        }
    }
    int rc = LRL_OK;
    // one arena, one set of per-sample cameras and of polled words per context: a render on another stream than the last one
    // waits (on the device) for that one to end
    if (c->renderedOnce && stream != c->lastRenderStream) CRUMB_TRY(c, "hipStreamWaitEvent(last render)", hipStreamWaitEvent(stream, c->evLastRender, 0));
    // scene changes still queued on the context's stream (lrl_water_step, lrl_scene_rebuild_bvh return before their kernels end)
    if (stream != c->stream) {
        CRUMB_TRY(c, "hipEventRecord(scene ready)", hipEventRecord(c->evSceneReady, c->stream));
        CRUMB_TRY(c, "hipStreamWaitEvent(scene ready)", hipStreamWaitEvent(stream, c->evSceneReady, 0));
    }
    // the per-sample cameras go to the device only when they changed: an unchanged camera costs no copy and no host wait,
    // so consecutive frames queue back to back
    if (c->transUploaded.size() != tr.size() || std::memcmp(c->transUploaded.data(), tr.data(), tr.size() * sizeof(LrlTrans)) != 0) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // renders in flight still read the old ones
        if (stream != c->stream) HIP_TRY(c, hipStreamSynchronize(stream));
        c->transUploaded.clear();
        if ((rc = ensure(c, c->transDev, tr.size() * sizeof(LrlTrans)))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->transDev.p, tr.data(), tr.size() * sizeof(LrlTrans), hipMemcpyHostToDevice, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));   // `tr` is a local
        c->transUploaded = tr;
    }
    unsigned long long* ctr = nullptr;
    if (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_COUNT_FETCHES)) {
        if ((rc = ensure(c, c->counterDev, LRL_CTR_COUNT * 8))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->counterDev.p, 0, LRL_CTR_COUNT * 8, stream));
        ctr = static_cast<unsigned long long*>(c->counterDev.p);
    }
    KScene k;
    fillKScene(c, k);
    KFrame f;
    f.trans = static_cast<const LrlTrans*>(c->transDev.p);
    f.width = c->width; f.height = c->height;
    f.rowBegin = p->rowBegin; f.rowEnd = p->rowBegin + rows;
    f.rows = rows;
    f.interleave = p->rowInterleave > 1 ? p->rowInterleave : 0u;
    f.shard = f.interleave ? p->rowShard : 0u;
    if (f.interleave) { f.rowBegin = 0; f.rowEnd = c->height; }
    f.alpha = float(s1 - s0) / float(spp);      // 1 exactly for a whole frame
    f.sampleBegin = s0; f.sampleEnd = s1; f.spp = spp;
    f.maxDepth = p->maxDepth;
    f.stackEntries = std::min(c->bvhDepth + 1, LRL_LDS_STACK_ENTRIES);
    f.timeline = nullptr;
    f.diffuse = (p->flags & LRL_FLAG_DIFFUSE_BOUNCE) ? 1u : 0u;
    if (f.diffuse && (p->flags & LRL_FLAG_MEGAKERNEL)) return fail(c, LRL_ERR_INVALID_ARG, "LRL_FLAG_DIFFUSE_BOUNCE is implemented in the wavefront pipeline only");
#ifdef LRL_DIAG
    const char* tlPath = std::getenv("LRL_TIMELINE");   // diagnostic builds: per-wave start/end ticks of this launch, dumped to the named file
#else
    const char* tlPath = nullptr;
#endif
    size_t tlWaves = 0;
    if (tlPath && *tlPath && (p->flags & LRL_FLAG_MEGAKERNEL)) {
        uint32_t tx = (c->width + LRL_TILE - 1) / LRL_TILE, ty = (rows + LRL_TILE - 1) / LRL_TILE;
        tlWaves = size_t(tx) * ty * (LRL_BLOCK_THREADS / 64);
        if ((rc = ensure(c, c->timelineDev, tlWaves * 8 * LRL_TIMELINE_WORDS))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->timelineDev.p, 0, tlWaves * 8 * LRL_TIMELINE_WORDS, stream));
        f.timeline = static_cast<unsigned long long*>(c->timelineDev.p);
    }
    const bool mega = (p->flags & LRL_FLAG_MEGAKERNEL) != 0;
    const bool perKernel = !mega && c->timing && (p->flags & LRL_FLAG_TIME_KERNELS);
    hipEvent_t e0 = c->evStart, e1 = c->evStop;
    if (c->timing && !perKernel && c->timingUsed < LRL_MAX_TIMED_LAUNCHES) {
        if (c->timingUsed == c->timingEvents.size()) {
            hipEvent_t a, b;
            HIP_TRY(c, hipEventCreate(&a));
            HIP_TRY(c, hipEventCreate(&b));
            c->timingEvents.emplace_back(a, b);
        }
        e0 = c->timingEvents[c->timingUsed].first; e1 = c->timingEvents[c->timingUsed].second;
        ++c->timingUsed;
    }
    auto finish = [&]() -> int {
        CRUMB_TRY(c, "hipEventRecord(last render)", hipEventRecord(c->evLastRender, stream));
        c->lastRenderStream = stream;
        c->renderedOnce = true;
        return LRL_OK;
    };
    if (!mega) {
        if ((rc = renderWavefront(c, p, k, f, rgbaDev, hitDev, ctr, stream, rows, e0, e1))) return rc;
        c->timed = (e0 == c->evStart);
        return finish();
    }
    HIP_TRY(c, hipEventRecord(e0, stream));
    if (p->mode == LRL_MODE_FAST)
        HIP_TRY(c, lrl_launch_render_fast(k, f, reinterpret_cast<float4*>(rgbaDev), reinterpret_cast<uint2*>(hitDev), ctr, p->variant == LRL_VARIANT_POOL, stream));
    else
        HIP_TRY(c, lrl_launch_render_parity(k, f, reinterpret_cast<float4*>(rgbaDev), reinterpret_cast<uint2*>(hitDev), ctr, p->variant == LRL_VARIANT_POOL, stream));
    HIP_TRY(c, hipEventRecord(e1, stream));
    c->timed = (e0 == c->evStart);
    if (f.timeline) {
        std::vector<unsigned long long> h(tlWaves * LRL_TIMELINE_WORDS);
        HIP_TRY(c, hipStreamSynchronize(stream));
        HIP_TRY(c, hipMemcpy(h.data(), c->timelineDev.p, tlWaves * 8 * LRL_TIMELINE_WORDS, hipMemcpyDeviceToHost));
        if (FILE* fp = std::fopen(tlPath, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    }
    return finish();
}

static int checkRender(LrlContext* c, const LrlRenderParams* p, uint32_t& rows) {
    if (!c || !p) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene || !c->haveCamera) return fail(c, LRL_ERR_NO_SCENE, "scene and camera must be set before rendering");
    if (p->maxDepth < 1 || p->maxDepth > LRL_MAX_RAY_DEPTH) return fail(c, LRL_ERR_LIMIT, "maxDepth must be 1..LRL_MAX_RAY_DEPTH");
    if (p->variant > LRL_VARIANT_POOL || p->mode > LRL_MODE_FAST) return fail(c, LRL_ERR_INVALID_ARG, "unknown variant / mode");
    if ((p->flags & LRL_FLAG_COUNT_FETCHES) && (p->mode != LRL_MODE_FAST || (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_MEGAKERNEL))))
        return fail(c, LRL_ERR_INVALID_ARG, "LRL_FLAG_COUNT_FETCHES goes with LRL_MODE_FAST, without LRL_FLAG_COUNTERS / LRL_FLAG_MEGAKERNEL");
    if (p->spp < 1 || p->spp > LRL_MAX_SPP) return fail(c, LRL_ERR_LIMIT, "spp must be 1..LRL_MAX_SPP");
    uint32_t r0 = p->rowBegin, r1 = p->rowEnd;
    if (r0 == 0 && r1 == 0) r1 = c->height;
    if (r0 >= r1 || r1 > c->height) return fail(c, LRL_ERR_INVALID_ARG, "row range invalid");
    if (p->rowInterleave > 1) {
        if (p->rowBegin || p->rowEnd) return fail(c, LRL_ERR_INVALID_ARG, "rowInterleave goes with rowBegin = rowEnd = 0");
        if (p->rowShard >= p->rowInterleave) return fail(c, LRL_ERR_INVALID_ARG, "rowShard must be below rowInterleave");
        if (p->flags & LRL_FLAG_MEGAKERNEL) return fail(c, LRL_ERR_INVALID_ARG, "rowInterleave is implemented in the wavefront pipeline only");
        r0 = 0;
        r1 = lrl_shard_rows(c->height, p->rowInterleave, p->rowShard);
        if (r1 == 0) return fail(c, LRL_ERR_INVALID_ARG, "this shard holds no rows of the frame (more shards than stripes)");
    }
    if (!(p->sampleBegin == 0 && p->sampleEnd == 0) && (p->sampleBegin >= p->sampleEnd || p->sampleEnd > p->spp))
        return fail(c, LRL_ERR_INVALID_ARG, "sample range invalid");
    rows = r1 - r0;
    return LRL_OK;
}

}  // namespace lrl_host

using namespace lrl_host;

extern "C" {

uint32_t lrl_shard_rows(uint32_t height, uint32_t interleave, uint32_t shard) {
    if (interleave <= 1) return shard == 0 ? height : 0u;
    if (shard >= interleave) return 0u;
    const uint32_t stripes = (height + LRL_STRIPE_ROWS - 1) / LRL_STRIPE_ROWS;
    if (shard >= stripes) return 0u;
    const uint32_t mine = (stripes - shard + interleave - 1) / interleave;      // stripes shard, shard + N, ...
    const uint32_t last = shard + (mine - 1) * interleave;                      // this shard's last stripe
    uint32_t rows = mine * LRL_STRIPE_ROWS;
    if (last == stripes - 1) rows -= stripes * LRL_STRIPE_ROWS - height;       // the frame's last stripe may be short
    return rows;
}

int lrl_render_async(LrlContext* c, const LrlRenderParams* p, float* rgbaDev, uint32_t* hitDev, void* hipStream) {
    uint32_t rows = 0;
    int rc = checkRender(c, p, rows);
    if (rc) return rc;
    if (!rgbaDev) return fail(c, LRL_ERR_INVALID_ARG, "rgba_out is null");
    HIP_TRY(c, hipSetDevice(c->device));
    return renderImpl(c, p, rgbaDev, hitDev, hipStream ? static_cast<hipStream_t>(hipStream) : c->stream, rows);
}

int lrl_render(LrlContext* c, const LrlRenderParams* p, float* rgba, uint32_t* hit) {
    uint32_t rows = 0;
    int rc = checkRender(c, p, rows);
    if (rc) return rc;
    if (!rgba) return fail(c, LRL_ERR_INVALID_ARG, "rgba_out is null");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool dev = (p->flags & LRL_FLAG_OUT_DEVICE) != 0;
    const size_t px = size_t(rows) * c->width;
    float* o = rgba;
    uint32_t* h = hit;
    if (!dev) {
        if ((rc = ensure(c, c->outDev, px * 16))) return rc;
        o = static_cast<float*>(c->outDev.p);
        if (hit) {
            if ((rc = ensure(c, c->hitDev, px * 8))) return rc;
            h = static_cast<uint32_t*>(c->hitDev.p);
        }
    }
    rc = renderImpl(c, p, o, h, c->stream, rows);
    if (rc) return rc;
    if (!dev) {
        HIP_TRY(c, hipMemcpyAsync(rgba, o, px * 16, hipMemcpyDeviceToHost, c->stream));
        if (hit) HIP_TRY(c, hipMemcpyAsync(hit, h, px * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_COUNT_FETCHES)) {
        unsigned long long v[LRL_CTR_COUNT];
        HIP_TRY(c, hipMemcpy(v, c->counterDev.p, sizeof(v), hipMemcpyDeviceToHost));
        c->counters.raysPrimary = v[LRL_CTR_PRIMARY]; c->counters.raysReflect = v[LRL_CTR_REFLECT];
        c->counters.raysRefract = v[LRL_CTR_REFRACT]; c->counters.raysShadow = v[LRL_CTR_SHADOW];
        c->counters.nodeVisits = v[LRL_CTR_NODES]; c->counters.triTests = v[LRL_CTR_TRI]; c->counters.sphereTests = v[LRL_CTR_SPH];
        c->counters.otherTests = v[LRL_CTR_OTHER]; c->counters.primaryHits = v[LRL_CTR_PRIMARY_HITS];
        c->counters.recordFetches = v[LRL_CTR_FETCH]; c->counters.laneTrips = v[LRL_CTR_LANE_TRIPS]; c->counters.waveTrips = v[LRL_CTR_WAVE_TRIPS]; c->counters.groupLanes = v[LRL_CTR_GROUP_LANES]; c->counters.groupRounds = v[LRL_CTR_GROUP_ROUNDS];
    }
    return LRL_OK;
}

}  // extern "C"
