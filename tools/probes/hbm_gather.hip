// Probe: what does HBM give a traversal-shaped access stream?  (the C5 roofline)
// A table far beyond every cache (4 GiB); every lane fetches whole records of it at random, either independently (as many
// fetches in flight as the hardware takes) or as a dependent chain (the next record's index comes out of the record: one fetch
// in flight per lane, the shape of a one-lane-per-ray BVH walk).  Known byte counts: under `rocprofv3 --pmc FETCH_SIZE` the same
// run calibrates the counter for this access shape (the guide calibrates it for 16-B-per-lane streaming only).
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_gather hbm_gather.hip ; run: ./hbm_gather [table MiB, a power of two; default 4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void fillKernel(uint4* table, size_t quads) {
    for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < quads; i += size_t(gridDim.x) * 256) {
        const uint32_t h = mix(uint32_t(i) * 2654435761u + uint32_t(i >> 32));
        table[i] = make_uint4(h, mix(h), mix(h + 1u), mix(h + 2u));
    }
}

// REC = record size in bytes (64 or 128); CHAIN: the next index depends on the record just read
template <int REC, bool CHAIN>
__global__ __launch_bounds__(256) void gatherKernel(const uint4* __restrict__ table, uint32_t recordMask, uint32_t steps, uint32_t* out) {
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 12345u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t r = s & recordMask;
        const uint4* p = table + size_t(r) * (REC / 16);
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
        if (REC == 128) {
            uint4 e = p[4], f = p[5], g = p[6], h = p[7];
            v ^= e.x ^ f.y ^ g.z ^ h.w;
        }
        acc += v;
        s = CHAIN ? mix(v + it) : mix(s + 0x9e3779b9u);
    }
    if (REC == 64) atomicAdd(out + 1, acc);      // checksum: the cooperative variants must arrive at the same
    if (acc == 0x12345678u) out[0] = acc;
}

// every lane its own REC-byte record with REC / 16 loads (16, 32 B): how does the rate scale with the loads per record?
template <int REC>
__global__ __launch_bounds__(256) void gatherSmallKernel(const uint4* __restrict__ table, uint32_t recordMask, uint32_t steps, uint32_t* out) {
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 12345u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t r = s & recordMask;
        const uint4* p = table + size_t(r) * (REC / 16);
        uint4 a = p[0];
        uint32_t v = a.x ^ a.w;
        if (REC == 32) { uint4 b = p[1]; v ^= b.y; }
        acc += v;
        s = mix(v + it);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// quad-cooperative: the four lanes of a quad fetch the four 16-B parts of ONE 64-B record per load instruction (four instructions
// = the quad's four records), so that a load instruction of the wave touches 16 records (pages) instead of 64; the parts come
// back to their owner through four quad-permute moves per register
template <bool CHAIN>
__global__ __launch_bounds__(256) void gatherQuadKernel(const uint4* __restrict__ table, uint32_t recordMask, uint32_t steps, uint32_t* out) {
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 12345u);
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u, part = lane & 3u;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t r = s & recordMask;
        uint4 got[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rj = uint32_t(__shfl(int(r), int((lane & ~3u) + j), 64));      // the record lane 4k+j wants
            got[j] = table[size_t(rj) * 4 + part];
        }
        // lane 4k+j now holds part (lane&3) of records 0..3 of its quad; it needs parts 0..3 of record j: a 4 x 4 transpose.
        // The probe only needs one word of each part: x of part 0, y of part 1, z of part 2, w of part 3 of MY record.
        uint32_t v = 0;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            // from lane 4k+pp, its got[my index].component(pp)
            uint32_t w0 = pp == 0 ? got[0].x : pp == 1 ? got[0].y : pp == 2 ? got[0].z : got[0].w;
            uint32_t w1 = pp == 0 ? got[1].x : pp == 1 ? got[1].y : pp == 2 ? got[1].z : got[1].w;
            uint32_t w2 = pp == 0 ? got[2].x : pp == 1 ? got[2].y : pp == 2 ? got[2].z : got[2].w;
            uint32_t w3 = pp == 0 ? got[3].x : pp == 1 ? got[3].y : pp == 2 ? got[3].z : got[3].w;
            const uint32_t src = (lane & ~3u) + pp;
            const uint32_t a0 = uint32_t(__shfl(int(w0), int(src), 64)), a1 = uint32_t(__shfl(int(w1), int(src), 64));
            const uint32_t a2 = uint32_t(__shfl(int(w2), int(src), 64)), a3 = uint32_t(__shfl(int(w3), int(src), 64));
            v ^= part == 0 ? a0 : part == 1 ? a1 : part == 2 ? a2 : a3;
        }
        acc += v;
        s = CHAIN ? mix(v + it) : mix(s + 0x9e3779b9u);
    }
    atomicAdd(out + 1, acc);
    if (acc == 0x12345678u) out[0] = acc;
}

// row-cooperative: lanes l, l + 16, l + 32, l + 48 (one column of the wave's four 16-lane rows) fetch the four parts of one record
// per load; the 4 x 4 transpose between rows and registers is 16 v_permlane32_swap / v_permlane16_swap (gfx950), no selects
template <bool CHAIN>
__global__ __launch_bounds__(256) void gatherRowKernel(const uint4* __restrict__ table, uint32_t recordMask, uint32_t steps, uint32_t* out) {
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 12345u);
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u, row = lane >> 4, col = lane & 15u;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t r = s & recordMask;
        uint4 g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rj = uint32_t(__shfl(int(r), int(col + 16u * j), 64));      // the record the lane of row j in my column wants
            g[j] = table[size_t(rj) * 4 + row];
        }
#define SWAP32(A, B) { auto t_ = __builtin_amdgcn_permlane32_swap(A, B, false, false); A = t_[0]; B = t_[1]; }
#define SWAP16(A, B) { auto t_ = __builtin_amdgcn_permlane16_swap(A, B, false, false); A = t_[0]; B = t_[1]; }
#define XPOSE(C) SWAP32(g[0].C, g[2].C) SWAP32(g[1].C, g[3].C) SWAP16(g[0].C, g[1].C) SWAP16(g[2].C, g[3].C)
        XPOSE(x) XPOSE(y) XPOSE(z) XPOSE(w)
        const uint32_t v = g[0].x ^ g[1].y ^ g[2].z ^ g[3].w;
        acc += v;
        s = CHAIN ? mix(v + it) : mix(s + 0x9e3779b9u);
    }
    atomicAdd(out + 1, acc);
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename K>
static int runK(const char* name, K kernel, int rec, const uint4* table, uint32_t records, uint32_t* out, uint32_t blocksPerCU, uint32_t steps) {
    const uint32_t blocks = 256 * blocksPerCU;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, records - 1, 4u, out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, records - 1, steps, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    uint32_t sum[2] = {0, 0}; CHECK(hipMemcpy(sum, out, 8, hipMemcpyDeviceToHost)); CHECK(hipMemset(out, 0, 8));
    std::printf("[checksum %08x] ", sum[1]);
    const double fetches = double(blocks) * 256 * steps, bytes = fetches * rec;
    std::printf("%-44s %2u waves/SIMD  %9.0f fetches x %3d B = %7.3f GB  %8.3f ms  %7.1f GB/s  %6.2f G records/s\n", name, blocksPerCU, fetches, rec,
                bytes / 1e9, ms, bytes / (ms * 1e-3) / 1e9, fetches / (ms * 1e-3) / 1e9);
    return 0;
}

template <int REC, bool CHAIN>
static int run(const char* name, const uint4* table, uint32_t records, uint32_t* out, uint32_t blocksPerCU, uint32_t steps) {
    const uint32_t blocks = 256 * blocksPerCU;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gatherKernel<REC, CHAIN>), dim3(blocks), dim3(256), 0, 0, table, records - 1, 4u, out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gatherKernel<REC, CHAIN>), dim3(blocks), dim3(256), 0, 0, table, records - 1, steps, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    uint32_t sum[2] = {0, 0}; CHECK(hipMemcpy(sum, out, 8, hipMemcpyDeviceToHost)); CHECK(hipMemset(out, 0, 8));
    std::printf("[checksum %08x] ", sum[1]);
    const double fetches = double(blocks) * 256 * steps, bytes = fetches * REC;
    std::printf("%-44s %2u waves/SIMD  %9.0f fetches x %3d B = %7.3f GB  %8.3f ms  %7.1f GB/s  %6.2f us per step\n", name, blocksPerCU, fetches, REC,
                bytes / 1e9, ms, bytes / (ms * 1e-3) / 1e9, ms * 1e3 / steps);
    return 0;
}

int main(int argc, char** argv) {
    const size_t bytes = size_t(argc > 1 ? std::atoi(argv[1]) : 4096) << 20;      // table size in MiB (a power of two)
    std::printf("table: %zu MiB\n", bytes >> 20);
    uint4* table; uint32_t* out;
    const bool contiguous = argc > 2 && argv[2][0] == 'c';      // physically contiguous: larger page-table fragments, fewer TLB misses?
    if (contiguous) { CHECK(hipExtMallocWithFlags((void**)&table, bytes, hipDeviceMallocContiguous)); std::printf("hipDeviceMallocContiguous\n"); }
    else CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL(fillKernel, dim3(4096), dim3(256), 0, 0, table, bytes / 16);
    CHECK(hipDeviceSynchronize());
    const uint32_t rec64 = uint32_t(bytes / 64), rec128 = uint32_t(bytes / 128);
    for (uint32_t occ : {4u, 8u}) {      // 256-thread blocks per CU = waves per SIMD
        if (run<64, false>("independent 64-B records", table, rec64, out, occ, 64)) return 1;
        if (run<64, true>("dependent chain of 64-B records", table, rec64, out, occ, 64)) return 1;
        if (run<128, false>("independent 128-B records", table, rec128, out, occ, 64)) return 1;
        if (run<128, true>("dependent chain of 128-B records", table, rec128, out, occ, 64)) return 1;
        if (runK("chain of 64-B records, quad-cooperative", gatherQuadKernel<true>, 64, table, rec64, out, occ, 64)) return 1;
        if (runK("chain of 64-B records, row-cooperative", gatherRowKernel<true>, 64, table, rec64, out, occ, 64)) return 1;
        if (runK("chain of 32-B records (2 loads)", gatherSmallKernel<32>, 32, table, uint32_t(bytes / 32), out, occ, 64)) return 1;
        if (runK("chain of 16-B records (1 load)", gatherSmallKernel<16>, 16, table, uint32_t(bytes / 16), out, occ, 64)) return 1;
    }
    return 0;
}
