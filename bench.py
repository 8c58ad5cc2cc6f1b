#!/usr/bin/env python3
"""bench.py — Mrays/s of the hot path on the BASELINE.json headline configuration.

Default workload (BASELINE.json configs[1]): Stanford bunny x 2 midpoint subdivisions (1 240 960 triangles) + ground
disc + 1 point light, 1920x1080, 1 spp, 4 bounces.  A "step" is one frame: the wavefront pipeline (generate, then trace + shade per
ray segment) over every pixel, scene and camera already resident in HBM, image written to HBM.  rays = primary + reflection +
refraction + shadow segments (device counters of an un-timed counting render that walks the tree in the reference's order; equal to
the CPU oracle's counts, tests/).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4|c5] [--scaling weak|strong] [--mode fast|parity]
                  [--builder reference|median|device] [--tuning key=value,...]

Roofline (two records in the one JSON line):
  roofline     the timed workload.  C1-C4 touch a working set that sits in the 256 MiB Infinity Cache (C2: 199 MB), so no HBM number
               bounds them.  What does, by the counters (profiles/r03_c2_pmc.json), is VALU issue: `bound` = "valu", `achieved` = wave64
               VALU instructions of the trace kernels per second (PMC pass SQ_INSTS_VALU of the same kernels on the same workload, a
               count that does not depend on the run, over this run's trace-kernel time: HIP events around every trace launch on the
               stream they run on), `peak` = 1024 SIMDs x 2.4 GHz / 4 cycles per instruction.  Second: `l1_gather` = bytes the timed
               kernels fetch (64 B x their record fetches, counted by a counting instantiation of those same kernels) against the
               vector L1's gather rate for that access shape measured on this chip (tools/probes/l1_gather.hip: 14.7 TB/s; the guide
               has no L1 figure).  `traffic` = bytes beyond L2 from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/), `hbm_frac`
               = traffic / kernel time / 8 TB/s; `work` = SURVEY 8d's algorithmic bytes (reference-order visit counts).  Workloads
               without a PMC record of their own (other scenes, --tuning) report `bound` = "l1" with the gather figure.
  roofline_c5  SURVEY 8d's C5, the configuration the memory roofline is judged on (100.5 M triangles, device-built tree, diffuse
               bounces, 16 spp; 22 GB of scene, far beyond the caches): `achieved` = the HBM traffic of the trace kernels (PMC passes of
               the same kernels on the same workload, profiles/r03_traffic_c5.json) / this run's trace-kernel time against the 8 TB/s
               HBM peak; `algorithmic_frac` = SURVEY 8d's algorithmic bytes over the same time (the caches serve part of them, so it
               can pass 1); the HBM line-request rate beside what tools/probes/hbm_gather.hip measured this chip to give such a
               stream, and valu_issue; `verified`: that scene's primary hits against a tree-free brute-force oracle on 256 pixels and
               FAST against PARITY over the frame.  N = 1 default runs only (--no-c5 skips it).
  Both records name the PMC summary they take their numerator from and say whether it was measured on the sources this library was
  built from (`profile_matches_build`: the summaries carry a hash of raytracing_amd/csrc); when it was not, `frac` is withheld.

N > 1 (one rank per GPU; `python bench.py --gpus N` starts the ranks itself through torch.distributed.run when it is not already
under one): the scene is replicated.
  --scaling weak   (default for c1/c2/c3/c5) every rank renders the whole frame for its own sample indices of an N x spp jitter table
                   (per-GPU work fixed); the partial means are summed onto rank 0 with one RCCL reduce per step inside the timed region, which
                   overlaps the rendering of the next frame (two image buffers, stream-level waits only).
  --scaling strong (default for c4: ONE 3840x2160 x 256 spp frame of the C2 scene) every rank renders its interleaved 16-row stripes
                   (LrlRenderParams.rowInterleave, one call), one RCCL all-gather per frame inside the timed region + the un-interleave.
value = rays of all ranks / max-over-ranks time.  The line says which collective library ran and how many ranks it saw.
  --rehearse       N ranks on GPU 0 with gloo on host copies (RCCL refuses two ranks on one device): the loop, not the library.
  --one-rank-rccl  with --gpus 1: a process group of one rank on RCCL and the N-rank step: the library's calls, not the traffic.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# vector-L1 gather rate for this kernel's access shape (every lane its own 64-B record, four dwordx4 loads): 171 CU-cycles per
# wave-round from L1 (tools/probes/l1_gather.hip on MI355X, gpurun_out/l1_gather.txt) = 4096 B / 171 cycles x 256 CUs x 2.4 GHz
L1_GATHER_PEAK_GBPS = 4096.0 / 171.2 * 256 * 2.4
INFINITY_CACHE_BYTES = 256 << 20

WORKLOADS = {
    #        width height depth spp  variant grid diffuse builder      scaling
    "c1": (1024, 1024, 10, 1, "bvh", 1, False, "reference", "weak"),
    "c2": (1920, 1080, 4, 1, "bvh", 1, False, "reference", "weak"),
    "c3": (1920, 1080, 8, 64, "pool", 1, False, "reference", "weak"),
    "c4": (3840, 2160, 4, 256, "bvh", 1, False, "reference", "strong"),
    "c5": (1920, 1080, 4, 16, "bvh", 9, True, "device", "weak"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = the workload's own (50 for c1/c2, 5 for c3/c5, 3 for c4)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1 = the workload's own")
    ap.add_argument("--mode", default="fast", choices=["fast", "parity"])
    ap.add_argument("--builder", default="", choices=["", "reference", "median", "sah", "device"],
                    help="who builds the BVH: the restated reference builder (host), a median-split host builder, or the library on the device")
    ap.add_argument("--subdivisions", type=int, default=2)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0, help="RayTraceDepth (0 = the workload's own)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 (default, the headline metric's configuration) | c1: the reference's default BVH scene as shipped, 1024 x 1024 | "
                         "c3: Pool demo scene, 64 spp | c4: the c2 scene at 3840 x 2160 x 256 spp | c5: 100.5 M triangles, diffuse bounces, 16 spp")
    ap.add_argument("--grid", type=int, default=0, help="mesh scenes: grid x grid replicated copies of the mesh (9 -> 100.5 M triangles)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (0 = the workload's own)")
    ap.add_argument("--diffuse", action="store_true", help="LRL_FLAG_DIFFUSE_BOUNCE: cosine-weighted continuation rays (c5 has it on)")
    ap.add_argument("--scaling", default="", choices=["", "weak", "strong"])
    ap.add_argument("--tuning", default="", help="LrlTuning fields, e.g. batchSlots=4194304,adapt=0,handoverList=48:48:32:16")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1, weak: the all-reduce of a frame does not overlap the next frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the roofline_c5 record of the default run")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = all)")
    ap.add_argument("--one-rank-rccl", action="store_true",
                    help="with --gpus 1: a process group of ONE rank on RCCL, and the step the N-rank run takes (render on a side stream, "
                         "dist.reduce(async_op=True) overlapping the next frame, or all_gather_into_tensor + un-interleave): the measured "
                         "configuration's own calls, on the one GPU a test has")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box (never the measured configuration): every rank on GPU 0, gloo instead of RCCL (which refuses two ranks on one device)")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start N ranks with torch.distributed.run as CHILD processes.  Nothing in this
    process has touched the GPU (no exec of a process that has)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def apply_tuning(tr, spec):
    if not spec:
        return
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        if k == "handoverList":
            kw[k] = [int(x) for x in v.split(":")]
        elif k == "workLow":
            kw[k] = float(v)
        else:
            kw[k] = int(v)
    tr.set_tuning(**kw)


def csrc_hash():
    """sha256 over the kernel and host sources the library is built from (raytracing_amd/csrc: *.hip, *.inc, *.h, Makefile), in name
    order.  The PMC summaries under profiles/ carry the hash of the sources they were measured on (tools/summarize_pmc.py); a count
    from other kernels than the ones being timed must not become this run's roofline numerator."""
    import hashlib
    d = os.path.join(REPO, "raytracing_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".inc", ".h")) or n == "Makefile":
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


PROFILE_ROUNDS = ("r03", "r02")      # newest first: the PMC record of the round whose kernels these are


def load_traffic(suffix, match):
    """PMC passes are separate runs (tools/run_pmc_tag.sh); their per-frame counts live in profiles/<round>_traffic<suffix>.json.
    Returns (HBM bytes of the trace kernels per frame, VALU wave-instructions of the trace kernels per frame or None, record): the
    record says which file, whether the sources it was measured on are the sources of this build (`profile_matches_build`) and with
    which LrlTuning the pass ran."""
    for rnd in PROFILE_ROUNDS:
        name = f"{rnd}_traffic{suffix}.json"
        path = os.path.join(REPO, "profiles", name)
        if not os.path.exists(path):
            continue
        tj = json.load(open(path))
        if any(tj.get(k) != v for k, v in match.items()):
            continue
        valu = None
        src = os.path.join(REPO, tj.get("source", ""))
        if os.path.isfile(src):
            valu = json.load(open(src)).get("trace_kernels_per_frame", {}).get("sq", {}).get("SQ_INSTS_VALU")
        rec = {"file": "profiles/" + name, "source": tj.get("source"), "csrc_sha256": tj.get("csrc_sha256"), "tuning": tj.get("tuning"),
               "profile_matches_build": tj.get("csrc_sha256") == csrc_hash()}
        return tj.get("hbm_bytes_per_frame"), valu, rec
    return None, None, None


# tools/probes/hbm_gather.hip on this chip (profiles/r02_hbm_gather_probe.txt): what the memory system gives a traversal-shaped stream
# of whole-record fetches, every one an L2 miss (one 128-B fabric request each)
GATHER_PROBE = {"source": "profiles/r02_hbm_gather_probe.txt",
                "requests_per_s_table_up_to_2GiB": 53e9, "requests_per_s_random_64B_records_16GiB": 19.6e9,
                "requests_per_s_16GiB_four_lanes_per_record": 45e9,
                "note": "beyond ~2 GiB a uniformly random gather is bound by address translation per lane, not by HBM; 53 G requests/s x 128 B = 6.8 TB/s"}
SIMDS, CLOCK_GHZ = 1024, 2.4          # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
FP32_VECTOR_SPEC_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: FP32 vector peak (reached only with packed v_pk_fma_f32: 2 x 64 lanes x 2 flop / 4 cycles)


class Workload:
    """One scene + camera + render parameters on one tracer context."""

    def __init__(self, a, name, local_rank, world, rank, scaling):
        from raytracing_amd import abi, host, scenes, tracer
        W, H, depth, spp, variant, grid, diffuse, builder, _ = WORKLOADS[name]
        self.name = name
        self.W, self.H = a.width or W, a.height or H
        self.depth = a.depth or depth
        self.spp = a.spp or spp
        self.grid = a.grid or grid
        self.diffuse = a.diffuse or diffuse
        self.builder = a.builder or builder
        self.subdivisions = a.subdivisions
        self.variant = abi.VARIANT_POOL if variant == "pool" else abi.VARIANT_BVH
        self.mode = abi.MODE_FAST if a.mode == "fast" else abi.MODE_PARITY
        self.mode_name = a.mode
        self.world, self.rank, self.scaling = world, rank, scaling
        t0 = time.time()
        if name == "c3":
            m = scenes.pool(builder="reference" if self.builder == "device" else self.builder)
        elif name == "c1":
            m = scenes.bvh_default(builder="reference" if self.builder == "device" else self.builder)
        else:
            m = scenes.mesh_scene(subdivisions=a.subdivisions, builder="none" if self.builder == "device" else self.builder, grid=self.grid)
        self.build_s = time.time() - t0
        self.model = m
        self.trans = m.camera(self.H)
        self.view = m.scene_view()
        if self.builder == "device":
            self.view.bvh = None
            self.view.bvhNodeNum = 0
        self.tr = tracer.Tracer(local_rank)
        apply_tuning(self.tr, a.tuning)
        t0 = time.time()
        self.tr.scene_upload(self.view)
        self.upload_s = time.time() - t0
        self.tr.set_camera(self.trans, self.W, self.H)
        self.tr.set_decay_origin((0, 0, 0, 1))
        self.info = self.tr.scene_info()
        # weak: every rank renders spp samples of a (spp * world)-entry jitter table, the whole frame
        # strong: every rank renders all spp samples of its interleaved stripes
        if scaling == "weak":
            total = self.spp * world
            self.jitter = host.jitter_table(total) if total > 1 else None
            self.samples = (rank * self.spp, (rank + 1) * self.spp) if world > 1 else None
            self.rows = None
        else:
            self.jitter = host.jitter_table(self.spp) if self.spp > 1 else None
            self.samples = None
            self.rows = ("interleave", world, rank) if world > 1 else None
        self.out_rows = self.tr.shard_rows(world, rank) if self.rows else self.H

    def kw(self):
        return dict(max_depth=self.depth, variant=self.variant, mode=self.mode, jitter=self.jitter, samples=self.samples, rows=self.rows,
                    diffuse=self.diffuse)

    def count(self):
        """Un-timed counting renders of this rank's share: rays + algorithmic bytes (reference visit order), and the record fetches of
        the FAST kernels themselves."""
        tr = self.tr
        tr.render(counters=True, want_hits=False, **self.kw())
        cn = tr.counters()
        self.cn = cn
        self.rays, self.alg_bytes = cn.rays, cn.algorithmic_bytes()
        self.fetch_bytes = None
        self.useful_lane_frac = None
        if self.mode_name == "fast":
            tr.render(count_fetches=True, want_hits=False, **self.kw())
            cf = tr.counters()
            self.fetch_bytes = 64 * int(cf.recordFetches)
            if cf.waveTrips:
                # lanes that carried a live ray (lean kernel) or a frontier entry (hand-over kernel), over 64 x the trips of their waves
                self.useful_lane_frac = float(cf.laneTrips) / (64.0 * float(cf.waveTrips))
                self.wave_trips = int(cf.waveTrips)
                # the two kernels apart: the one-lane-per-ray kernel (lanes whose ray is still going) and the hand-over kernel (lanes with a frontier entry)
                lw, ll = int(cf.waveTrips) - int(cf.groupRounds), int(cf.laneTrips) - int(cf.groupLanes)
                self.lane_use = {"lean_wave_trips": lw, "lean_useful_lane_frac": round(ll / (64.0 * lw), 4) if lw else None,
                                 "handover_rounds": int(cf.groupRounds), "handover_useful_lane_frac": round(int(cf.groupLanes) / (64.0 * int(cf.groupRounds)), 4) if cf.groupRounds else None}

    def render_into(self, ptr, stream=None, time_kernels=False):
        self.tr.render_device(ptr, stream=stream, time_kernels=time_kernels, **self.kw())

    def trace_kernel_ms(self, ptr, frames):
        """Device time of the trace launches of one frame: HIP events around every trace launch (a separate short pass: the event
        records add gaps, so it is not the pass `value` comes from).  The library keeps 4096 event pairs."""
        self.tr.timing_begin()
        for _ in range(frames):
            self.render_into(ptr, time_kernels=True)
        self.tr.sync()
        ms, launches = self.tr.timing_end()
        return ms / frames, launches // frames

    def describe(self):
        n = self.view.num.triangleNum
        if self.name == "c1":
            s = f"C1: the reference's default BVH scene ({n} triangles + spheres, discs, cylinders, cones), "
        elif self.name == "c3":
            s = f"C3: Pool demo scene ({n} triangles, water at rest), Pool shader variant, "
        else:
            s = (f"{self.name.upper()}: Stanford bunny x{self.subdivisions} subdivisions"
                 f"{f', {self.grid}x{self.grid} copies' if self.grid > 1 else ''} ({n} triangles) + ground disc + 1 light, ")
        s += f"{self.W}x{self.H}, {self.spp} spp" + (" per GPU" if self.scaling == "weak" and self.world > 1 else "") + f", RayTraceDepth {self.depth}"
        return s + (", diffuse bounces (Philox, cosine-weighted)" if self.diffuse else "")

    def working_set(self):
        d = self.tr.scene_device_bytes()
        # a FAST frame touches the four-wide nodes in use, one copy of the hot triangle records and the cold ones of the hits
        return self.info["wideNodes"] * 64 + self.view.num.triangleNum * 64 + d["tri_cold"] + d["other"]

    def roofline(self, kernel_ms, launches, frame_ms, traffic, valu=None, profile=None):
        resident = self.working_set() <= INFINITY_CACHE_BYTES
        stale = profile is not None and not profile["profile_matches_build"]
        r = {"kernel": ("trace kernels of one frame (wfTraceLeanKernel + the hand-over kernel: wfTraceGroupKernel, or wfTraceSwarmKernel beyond the caches; where a material transmits also wfTailKernel, the launch that finishes a pass from its sixth segment on and shades as well)" if self.mode_name == "fast"
                        else "trace kernels of one frame (wfTraceKernel)"),
             "kernel_ms": round(kernel_ms, 4), "launches_per_frame": launches, "frame_ms": round(frame_ms, 4),
             "working_set_bytes": self.working_set(), "traffic": traffic,
             "hbm_frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic else None,
             "work": {"algorithmic_bytes_per_frame": int(self.alg_bytes),
                      "algorithmic_GBps": round(self.alg_bytes / (kernel_ms * 1e-3) / 1e9, 2),
                      "counts": {"node_visits": int(self.cn.nodeVisits), "tri_tests": int(self.cn.triTests), "sphere_tests": int(self.cn.sphereTests),
                                 "other_tests": int(self.cn.otherTests), "rays": int(self.rays)}}}
        if valu:
            # VALU issue: PMC pass SQ_INSTS_VALU of the same kernels on the same workload (a count that does not depend on the run: profiles/)
            # over this run's kernel time; a wave64 instruction occupies its 16-lane SIMD for four cycles
            peak = SIMDS * CLOCK_GHZ / 4.0                     # G wave-instructions per second
            spec = FP32_VECTOR_SPEC_TFLOPS * 1e3 / 128.0       # G wave-instructions/s if every one were a wave64 fma at the guide's FP32 vector peak (128 flop each)
            busy = valu / (kernel_ms * 1e-3) / 1e9 / peak
            r["valu_issue"] = {"wave_instructions_per_frame": int(valu), "achieved": round(valu / (kernel_ms * 1e-3) / 1e9, 2), "peak": round(peak, 1),
                               "unit": "G wave-instructions/s", "busy_frac": round(busy, 4),
                               "frac_vs_fp32_spec": round(valu / (kernel_ms * 1e-3) / 1e9 / spec, 4), "fp32_spec_peak": round(spec, 1),
                               "useful_lane_frac": round(self.useful_lane_frac, 4) if self.useful_lane_frac else None,
                               "useful_busy_frac": round(busy * self.useful_lane_frac, 4) if self.useful_lane_frac else None,
                               "lane_use": getattr(self, "lane_use", None),
                               "useful_note": "useful_lane_frac = lanes that carried a live ray (a frontier entry in the hand-over kernel) / (64 x wave-trips), counted by the counting "
                                              "instantiation of the timed kernels on this run's frame; a wave lives until its longest ray is done, and the idle lanes' instructions are "
                                              "issued all the same: useful_busy_frac = busy_frac x useful_lane_frac.  frac_vs_fp32_spec prices the same count against the guide's "
                                              "157.3 TFLOP/s FP32 vector peak (packed fma: two wave64 fmas per 4 cycles), which unpacked code cannot reach",
                               "note": "rocprof's VALUBusy convention: 4 cycles per wave64 instruction (what SQ_ACTIVE_INST_VALU itself counts).  "
                                       "tools/probes/valu_rate.hip (profiles/r02_valu_rate_probe.txt): v_fma / v_max3 / v_cvt_f32_ubyte / v_cmp take 4.0-4.4 cycles, "
                                       "v_mul_f32 / v_mov_b32 2.4-2.6, v_rcp_f32 8.3, so the figure is an upper estimate of the pipe's occupancy (the Pool's "
                                       "mix reaches 1.19 by this convention)"}
        if resident and self.fetch_bytes:
            ach = self.fetch_bytes / (kernel_ms * 1e-3) / 1e9
            l1 = {"achieved": round(ach, 2), "peak": round(L1_GATHER_PEAK_GBPS, 1), "unit": "GB/s", "frac": round(ach / L1_GATHER_PEAK_GBPS, 5),
                  "fetched_bytes_per_frame": int(self.fetch_bytes),
                  "note": "64 B x record fetches of the timed kernels (their counting instantiation) against the vector L1's rate for 64 lanes that each read their "
                          "OWN record (tools/probes/l1_gather.hip); lanes that share a record cost less, so scenes of a few thousand triangles can pass 1"}
            if valu:
                # what binds a cache-resident frame: the counters say VALU issue (profiles/r02_c2_pmc.json), with the L1 gather rate second
                r.update({"bound": "valu", "level": "VALU issue of the trace kernels; the working set is resident in the Infinity Cache / L2, HBM is nearly idle",
                          "achieved": r["valu_issue"]["achieved"], "peak": r["valu_issue"]["peak"], "unit": r["valu_issue"]["unit"],
                          "frac": r["valu_issue"]["busy_frac"], "l1_gather": l1})
            else:
                r.update({"bound": "l1", "level": "vector L1 gather rate; the working set is resident in the Infinity Cache / L2, HBM is nearly idle"})
                r.update({k: l1[k] for k in ("achieved", "peak", "unit", "frac", "fetched_bytes_per_frame")})
                r["l1_note"] = l1["note"]
        else:
            # beyond the caches: the bytes that did cross the HBM interface (PMC passes of the same kernels on the same workload: FETCH_SIZE x 2 +
            # WRITE_SIZE, a count that barely depends on the run) over this run's kernel time; without a PMC record, SURVEY 8d's algorithmic
            # bytes (the caches serve part of those, so that figure can pass 1)
            alg = self.alg_bytes / (kernel_ms * 1e-3) / 1e9
            ach = traffic / (kernel_ms * 1e-3) / 1e9 if traffic else alg
            r.update({"bound": "hbm", "level": "HBM: the working set is far beyond the 256 MiB Infinity Cache" if not resident else "Infinity Cache resident",
                      "basis": "HBM traffic of the trace kernels measured by PMC (profiles/)" if traffic else "algorithmic bytes (SURVEY 8d)",
                      "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                      "algorithmic_frac": round(alg / HBM_PEAK_GBPS, 5),
                      "fetched_bytes_per_frame": int(self.fetch_bytes) if self.fetch_bytes else None})
        if traffic:
            # every L2 miss is one 128-B request (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ in the PMC passes)
            r["hbm_line_requests_per_s"] = round(traffic / 128.0 / (kernel_ms * 1e-3), 0)
            if not resident:
                r["gather_probe"] = GATHER_PROBE
        if profile is not None:
            r["profile"] = profile
            r["profile_matches_build"] = profile["profile_matches_build"]
            if stale:
                # the PMC counts were measured on other sources than the ones this build was made from: they say nothing about these kernels
                r["stale_frac"] = r.get("frac")
                r["frac"] = None
                r["frac_note"] = "profile_matches_build is false: the PMC numerator in profiles/ belongs to other kernels; frac withheld (stale_frac: what it would have been)"
                return r
        if not (0.0 < r["frac"] <= 1.0):
            r["frac_note"] = ("frac above 1: lanes in step share records, which the per-lane L1 gather peak does not know (l1_note)" if r["bound"] == "l1"
                              else "frac outside (0, 1]: the caches serve part of these bytes; see hbm_frac and work")
        return r


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a))

    import torch
    import torch.distributed as dist

    from raytracing_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearse = world > 1 and a.rehearse
    if rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(90, exit=True)     # a hang names its line
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or a.one_rank_rccl
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if rehearse else "nccl", **({} if rehearse else {"device_id": dev}))
    elif dist_on:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    ok = True
    try:
        run(a, torch, dist, sharding, rank, local_rank, world, dev, rehearse, dist_on)
    except BaseException:
        ok = False
        import traceback
        traceback.print_exc()
    if dist_on:
        if ok:
            dist.barrier()
            dist.destroy_process_group()
        else:
            os._exit(1)      # no collective error path: the launcher ends the other ranks
    elif not ok:
        sys.exit(1)


def run(a, torch, dist, sharding, rank, local_rank, world, dev, rehearse, dist_on):
    scaling = a.scaling or WORKLOADS[a.workload][8]
    steps = a.steps or {"c1": 50, "c2": 50, "c3": 5, "c4": 3, "c5": 5}[a.workload]
    # (at least eight: three frames settle the hand-over thresholds, four are timed by the library itself to decide on triangle trips)
    warmup = a.warmup if a.warmup >= 0 else {"c1": 10, "c2": 10, "c3": 8, "c4": 8, "c5": 8}[a.workload]
    wl = Workload(a, a.workload, local_rank, world, rank, scaling)
    wl.count()
    W, H = wl.W, wl.H

    stream = torch.cuda.Stream(device=dev) if dist_on else None
    raw = stream.cuda_stream if stream else None
    if scaling == "strong":
        index = sharding.interleave_index(H, world)
        idx_dev, pad = index[0].to(dev), index[1]
        part = torch.zeros((pad, W, 4), dtype=torch.float32, device=dev)
        gathered = torch.empty((world * pad, W, 4), dtype=torch.float32, device=dev) if dist_on else part
        frame = [None]
        outs = [part]
    else:
        outs = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2 if dist_on else 1)]
    works = [None, None]
    counter = [0]
    # The all-reduce of frame i overlaps the rendering of frame i + 1 (RCCL: stream-level waits only, the host never blocks).  Not in
    # the one-GPU rehearsal: there the collectives run on host copies.  What round 1 recorded as a stall "inside lrl_render_async" was
    # traced in round 2 (LrlTuning.debug = 2 prints a line around every HIP call of the render path; gpurun_out/r2_reh_weak*.err): every
    # call of the render path returns, and a render with nothing of gloo's queued before it finishes on the device; what stops is the
    # render stream's wait on the copy-back event of gloo's DEVICE-tensor all-reduce (its worker thread stages through pinned memory on
    # pool streams of its own), with two processes on one GPU.  RCCL has no such thread and no staging.
    overlap = dist_on and scaling == "weak" and not a.no_overlap and not rehearse

    # the frame is needed in one place (the reference shows it in one window): a reduce to rank 0 moves half the bytes of an all-reduce
    def all_reduce_(t):
        if rehearse:
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)

    def all_gather_(full, mine):
        if rehearse:
            h = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(h, mine.cpu())
            full.copy_(h)
        else:
            dist.all_gather_into_tensor(full, mine)

    def step():
        if not dist_on:
            wl.render_into(outs[0].data_ptr())
            if scaling == "strong":
                frame[0] = part.index_select(0, idx_dev)          # the same un-interleave the N-rank frame pays
            return
        with torch.cuda.stream(stream):
            if scaling == "strong":
                # one frame: my interleaved stripes, one all-gather, un-interleave
                wl.render_into(part.data_ptr(), stream=raw)
                all_gather_(gathered, part)
                frame[0] = gathered.index_select(0, idx_dev)
            elif not overlap:
                wl.render_into(outs[0].data_ptr(), stream=raw)
                all_reduce_(outs[0])
            else:
                # frame i renders (into the buffer whose all-reduce, of frame i - 2, was waited for in the last step) while the
                # all-reduce of frame i - 1 runs; then that one is waited for (a wait of the render stream, not of the host) and
                # the all-reduce of frame i is queued behind its render: never more than one collective in flight
                b = counter[0] & 1
                wl.render_into(outs[b].data_ptr(), stream=raw)
                if works[1 - b] is not None:
                    works[1 - b].wait()
                    works[1 - b] = None
                works[b] = dist.reduce(outs[b], dst=0, op=dist.ReduceOp.SUM, async_op=True)
        counter[0] += 1
        if rehearse:
            print(f"[rehearsal] rank {rank}: step {counter[0]} queued", file=sys.stderr, flush=True)

    def drain():
        for i in (0, 1):
            if works[i] is not None:
                works[i].wait()
                works[i] = None
        wl.tr.sync()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    drain()
    # The library times frames each way before it settles on triangle trips for a scene and frame shape (LrlRenderInfo.trialState = 1
    # while that is under way): the measured steps start only after the decision, so that they all run the same kernels.  Every rank
    # takes the same number of extra frames (the collectives pair up): a fixed count after which the state is asserted.
    extra = 0
    while wl.tr.last_render_info()["trialState"] == 1 and extra < 24:
        for _ in range(4):
            step()
        drain()
        extra += 4
    if dist_on:
        mx_ex = torch.tensor([extra], dtype=torch.int64, device="cpu" if rehearse else dev)
        dist.all_reduce(mx_ex, op=dist.ReduceOp.MAX)
        for _ in range(int(mx_ex[0]) - extra):
            step()
        drain()
        extra = int(mx_ex[0])
    settled = wl.tr.last_render_info()["trialState"] != 1
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wl.tr.timing_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frame_ms_total, frames = wl.tr.timing_end()

    scratch_out = outs[0].data_ptr()
    kernel_ms, launches = wl.trace_kernel_ms(scratch_out, 5 if wl.spp <= 16 else 1)
    rinfo = wl.tr.last_render_info()

    tot = torch.tensor([elapsed, float(wl.rays)], dtype=torch.float64, device=dev)
    per_rank_ms = [elapsed / steps * 1e3]
    backend = None
    if dist_on:
        backend = dist.get_backend()
        if rehearse:
            tot = tot.cpu()
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        every = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(every, tot)
        per_rank_ms = [float(e[0]) / steps * 1e3 for e in every]
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, rays_all = float(mx[0]), float(tot[1])
    else:
        rays_all = float(wl.rays)

    if rank != 0:
        wl.tr.close()
        return
    ms_per_step = elapsed / steps * 1e3
    value = rays_all * steps / elapsed / 1e6
    frame_ms = frame_ms_total / max(frames, 1)
    traffic = valu = profile = None
    # (weak scaling: rank 0's frame is the one-GPU frame with its own sample index, the same kernels over the same scene)
    if (world == 1 or scaling == "weak") and a.workload == "c2" and not a.tuning:
        traffic, valu, profile = load_traffic("", {"workload_triangles": int(wl.view.num.triangleNum), "mode": a.mode, "builder": wl.builder,
                                                   "width": W, "height": H, "spp": wl.spp, "depth": wl.depth, "diffuse": wl.diffuse})
    if world == 1 and a.workload == "c5" and not a.tuning and not (a.grid or a.spp or a.depth or a.width or a.builder or a.diffuse):
        traffic, valu, profile = load_traffic("_c5", {"workload_triangles": int(wl.view.num.triangleNum), "spp": wl.spp, "depth": wl.depth})
    if dist_on:
        # the collective library that actually ran: "nccl" is RCCL on ROCm; the one-GPU rehearsal runs gloo on host copies
        lib_name = {"nccl": "RCCL", "gloo": "gloo (host copies: one-GPU rehearsal, not the measured configuration)"}.get(backend, str(backend))
        par = (f"replicated scene, {wl.spp} sample indices per GPU, {lib_name} reduce of the image onto rank 0" + (" overlapping the next frame" if overlap else "")
               if scaling == "weak" else f"replicated scene, interleaved 16-row stripes per GPU, {lib_name} all-gather of the frame")
    else:
        par = "1 GPU"
    line = {
        "metric": "Mrays/s (whole node) at 1080p 4-bounce", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl.describe(), "bvh_builder": wl.builder, "bvh_nodes": wl.info["bvhNodes"], "bvh_depth": wl.info["bvhDepth"],
                   "bvh_wide_depth": wl.info["wideDepth"], "arithmetic": a.mode, "parallelism": par,
                   "rays_per_frame": int(rays_all), "host_scene_build_s": round(wl.build_s, 2), "scene_upload_s": round(wl.upload_s, 2),
                   "device_scene_bytes": wl.tr.scene_device_bytes(),
                   "collective_backend": backend, "ranks_seen_by_backend": dist.get_world_size() if dist_on else 1,
                   "ms_per_step_per_rank": [round(x, 4) for x in per_rank_ms],
                   "schedule": dict({k: rinfo[k] for k in ("passes", "batch", "iterations", "compactFrom", "groupLanes", "frontierEntries", "contSlots", "coopFetch", "streamWords", "phasedTrips", "handoverTrips", "trialState")},
                                    settled=bool(settled), extra_warmup_frames=extra, deterministic=int(wl.tr.get_tuning().deterministic))},
        "roofline": wl.roofline(kernel_ms, launches, frame_ms, traffic, valu, profile),
    }
    if not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(a, wl)
    wl.tr.close()
    if world == 1 and a.workload == "c2" and not a.no_c5 and not a.tuning and not a.grid:
        line["roofline_c5"] = roofline_c5(a, local_rank, torch, dev)
    print(json.dumps(line), flush=True)


def roofline_c5(a, local_rank, torch, dev):
    """SURVEY 8d's C5 on this GPU, timed by this run: the configuration the memory roofline is judged on."""
    import copy
    b = copy.copy(a)
    b.width = b.height = b.depth = b.spp = b.grid = 0
    b.builder, b.diffuse, b.tuning = "", False, ""
    wl = Workload(b, "c5", local_rank, 1, 0, "weak")
    wl.count()
    out = torch.empty((wl.H, wl.W, 4), dtype=torch.float32, device=dev)
    for _ in range(8):
        wl.render_into(out.data_ptr())
    wl.tr.sync()
    extra = 0
    while wl.tr.last_render_info()["trialState"] == 1 and extra < 24:      # the library's own timed trial ends before the timed steps begin
        wl.render_into(out.data_ptr())
        wl.tr.sync()
        extra += 1
    wl.tr.timing_begin()
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        wl.render_into(out.data_ptr())
    wl.tr.sync()
    elapsed = time.perf_counter() - t0
    frame_ms_total, frames = wl.tr.timing_end()
    kernel_ms, launches = wl.trace_kernel_ms(out.data_ptr(), 1)
    rinfo = wl.tr.last_render_info()
    traffic, valu, profile = load_traffic("_c5", {"workload_triangles": int(wl.view.num.triangleNum), "spp": wl.spp, "depth": wl.depth})
    r = wl.roofline(kernel_ms, launches, frame_ms_total / max(frames, 1), traffic, valu, profile)
    r.update({"workload": wl.describe(), "bvh_builder": wl.builder, "bvh_depth": wl.info["bvhDepth"], "bvh_wide_depth": wl.info["wideDepth"],
              "Mrays_per_s": round(wl.rays * steps / elapsed / 1e6, 2), "ms_per_frame": round(elapsed / steps * 1e3, 3), "steps": steps,
              "rays_per_frame": int(wl.rays), "host_scene_build_s": round(wl.build_s, 2), "scene_upload_and_build_s": round(wl.upload_s, 2),
              "device_scene_bytes": wl.tr.scene_device_bytes(),
              "schedule": dict({k: rinfo[k] for k in ("passes", "batch", "iterations", "compactFrom", "groupLanes", "frontierEntries", "phasedTrips", "handoverTrips", "trialState")},
                               extra_warmup_frames=extra)})
    if not a.no_cpu_baseline:
        r["verified"] = verify_c5(wl)
    wl.tr.close()
    return r


def verify_c5(wl):
    """The image behind roofline_c5, looked at: the 100.5 M-triangle scene is the only one whose byte offsets pass 2^32, and no host tree of
    it is affordable.  (i) A tree-free brute-force oracle (oracle_bruteforce_primary: every triangle against the primary rays of 256
    scattered pixels; part of the CPU-checker leg like cpu_baseline, never of a timed region) against the FAST and the PARITY primary
    ids of sample 0; (ii) FAST against PARITY ids over the whole frame; mismatches must be proven floating-point ties
    (oracle_api.tie_adjusted_match).  Path: /root/reference/BVH/shaders/RendererFragment0.cpp:406-632."""
    from oracle import oracle_api as O
    from raytracing_amd import abi, host
    t0 = time.time()
    tr0 = host.sample_trans(wl.trans, wl.jitter[:1]) if wl.jitter is not None else wl.trans
    kw = dict(wl.kw(), samples=(0, 1))
    _, fhits = wl.tr.render(**dict(kw, mode=abi.MODE_FAST))
    _, phits = wl.tr.render(**dict(kw, mode=abi.MODE_PARITY))
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    px = O.scattered_pixels(wl.W, wl.H, 256)
    bhits, _ = O.render_bruteforce(wl.view if wl.builder != "device" else wl.model.scene_view(), tr0, wl.W, wl.H, px, threads=threads)
    view = wl.model.scene_view()
    out = {"pixels": int(px.shape[0]), "pixels_on_triangles": int((bhits[:, 0] == 2).sum()), "oracle": "tree-free brute force, every triangle against every sampled primary ray"}
    for name, hits in (("fast", fhits), ("parity", phits)):
        share, exact, bad = O.tie_adjusted_match(hits[px[:, 1], px[:, 0]], bhits, view, tr0, wl.W, wl.H, pixels=px)
        out[f"ids_{name}_vs_bruteforce"] = {"tie_adjusted": share, "exact": exact, "not_ties": int(len(bad))}
    share, exact, bad = O.tie_adjusted_match(fhits, phits, view, tr0, wl.W, wl.H)
    out["ids_fast_vs_parity_whole_frame"] = {"tie_adjusted": round(share, 6), "exact": round(exact, 6), "not_ties": int(len(bad)), "pixels": int(wl.W * wl.H)}
    out["ids"] = min(out["ids_fast_vs_bruteforce"]["tie_adjusted"], out["ids_parity_vs_bruteforce"]["tie_adjusted"], share)
    out["ok"] = bool(out["ids"] >= 0.9999 and out["ids_fast_vs_bruteforce"]["not_ties"] == 0 and out["ids_parity_vs_bruteforce"]["not_ties"] == 0)
    out["seconds"] = round(time.time() - t0, 1)
    return out


def cpu_baseline(a, wl):
    """The CPU oracle (C++ restatement of the reference shader, OpenMP over rows) on a bounded stripe of the SAME frame, plus the
    recorded llvmpipe run of the reference's own shaders on C1 (it cannot run on the GPU box: /root/reference does not travel)."""
    from oracle import oracle_api as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, cores)      # the GPU box sets OMP_NUM_THREADS to its CPU share (16)
    if wl.builder == "device":
        if wl.view.num.triangleNum > 4_000_000:
            return {"value": None, "note": "skipped: the oracle needs a host-built tree, minutes of work at this size"}
        wl.model.data_init("median")          # the oracle walks a host-built tree: the image does not depend on the tree
    H, W = wl.H, wl.W
    rows = a.cpu_rows or H
    r0 = max(0, (H - rows) // 2)
    # repeat the stripe until ~12 s of CPU work have been timed (at least twice)
    rays, dt, reps = 0, 0.0, 0
    while dt < 12.0 or reps < 2:
        t0 = time.perf_counter()
        _, _, c = O.render(wl.view if wl.builder != "device" else wl.model.scene_view(), wl.trans, W, H, max_depth=wl.depth, variant=wl.variant,
                           rows=(r0, r0 + rows), threads=threads, want_hits=False, diffuse=wl.diffuse)
        dt += time.perf_counter() - t0
        rays += c.rays
        reps += 1
        if reps >= 200:
            break
    out = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "host_cpus_visible": cores, "kind": "port",
           "sample": f"rows {r0}..{r0 + rows} of the same {W}x{H} frame (one sample per pixel) rendered {reps} times ({rays} rays, {dt:.1f} s), "
                     f"oracle/liboracle.so, OpenMP over rows, {threads} threads"}
    p = os.path.join(REPO, "profiles", "r02_baseline_llvmpipe_c1.json")
    if os.path.exists(p):
        b1 = json.load(open(p))
        out["llvmpipe"] = {"note": "static record, measured once in the build container (tools/baseline_llvmpipe.py): the reference's own GLSL on "
                                   "Mesa llvmpipe, C1 (BASELINE configs[0]); not timed by this run",
                           "config": b1["config"], "host": b1["host"], "LP_NUM_THREADS": b1["LP_NUM_THREADS"], "runs": b1["runs"]}
    return out


if __name__ == "__main__":
    main()
