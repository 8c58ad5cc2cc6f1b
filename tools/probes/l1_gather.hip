// Probe: what does the vector L1 (TA/TCP/TD) charge for per-lane 16-byte loads, by access pattern?  
// Each wave runs ITER rounds of LOADS global_load_dwordx4 per lane on an L1/L2-resident table; the chip is filled with waves.
// build: hipcc --offload-arch=gfx950 -O3 -o l1_gather l1_gather.hip ; run: ./l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: every lane its own 64-B record, 4 loads (the lean trace kernel's pattern)
// mode 1: every lane its own 32-B record, 2 loads
// mode 2: all lanes the same 64-B record, 4 loads
// mode 3: quad-cooperative: lanes 4k..4k+3 load the four 16-B parts of one record per instruction; 4 instructions = 64 records
// mode 4: lanes 2k, 2k+1 share a 64-B record, 4 loads
// mode 5: every lane its own 16-B record, 1 load
template <int MODE>
__global__ __launch_bounds__(64) void probe(const float4* __restrict__ table, uint32_t records, uint32_t iters, float* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t s = blockIdx.x * 64u + lane;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        uint32_t r = (s >> 8) % records;
        if (MODE == 2) r = __builtin_amdgcn_readfirstlane(r);
        if (MODE == 4) r = __shfl(r, lane & ~1u, 64);
        if (MODE == 0 || MODE == 2 || MODE == 4) {
            const float4* p = table + size_t(r) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            const float4* p = table + size_t(r) * 2;
            const float4 a = p[0], b = p[1];
            acc += a.x + b.y;
        } else if (MODE == 5) {
            acc += table[r].x;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rj = __shfl(r, (lane & ~3u) + j, 64);      // record wanted by lane 4k+j
                acc += table[size_t(rj) * 4 + (lane & 3u)].x;
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE> static int run(const char* name, const float4* table, uint32_t records, float* out, int lines) {
    const uint32_t iters = 2000, blocks = 256 * 4 * 8 * 2;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, table, records, 10u, out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, table, records, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double waveRounds = double(blocks) * iters, cyc = ms * 1e-3 * 2.4e9 * 256 / waveRounds;
    std::printf("%-34s records %8u: %7.3f ms  -> %6.1f CU-cycles per wave-round (%d load instr), %5.2f per lane-line\n", name, records, ms, cyc, lines,
                cyc / 64.0);
    return 0;
}

int main() {
    const uint32_t maxRec = 1u << 21;      // 128 MB of 64-B records
    float4* table; float* out;
    CHECK(hipMalloc(&table, size_t(maxRec) * 64)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 0, size_t(maxRec) * 64));
    for (uint32_t records : {128u, 8192u, 1u << 21}) {       // 8 KB (L1), 512 KB (L2), 128 MB (Infinity Cache)
        if (run<0>("own 64-B record, 4 loads", table, records, out, 4)) return 1;
        if (run<1>("own 32-B record, 2 loads", table, records, out, 2)) return 1;
        if (run<5>("own 16-B record, 1 load", table, records, out, 1)) return 1;
        if (run<2>("same 64-B record, 4 loads", table, records, out, 4)) return 1;
        if (run<4>("pairs share a record, 4 loads", table, records, out, 4)) return 1;
        if (run<3>("quad-cooperative, 4 loads", table, records, out, 4)) return 1;
    }
    return 0;
}
