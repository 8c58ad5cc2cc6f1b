// Probe: how many cycles does a wave64 VALU instruction occupy its SIMD on MI355X?  (the peak of bench.py's valu_issue record)
// Every SIMD holds 8 waves that each issue a long run of independent instructions of one kind; rate = instructions / SIMD / cycle at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(X) X X X X X X X X
#define BODY(ASM)                                                                                   \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;     \
    const float b = 1.0001f, c = 0.5f;                                                              \
    for (uint32_t i = 0; i < iters; ++i) {                                                          \
        REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)   \
    }                                                                                               \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0;

#define FMA(K) "v_fma_f32 %" #K ", %" #K ", %8, %9\n"
#define MAX3(K) "v_max3_f32 %" #K ", %" #K ", %8, %9\n"
#define CVT(K) "v_cvt_f32_ubyte1 %" #K ", %" #K "\n"
#define CND(K) "v_cndmask_b32 %" #K ", %" #K ", %8, vcc\n"
#define MUL(K) "v_mul_f32 %" #K ", %" #K ", %8\n"
#define CMP(K) "v_cmp_lt_f32 vcc, %" #K ", %8\n"
#define RCP(K) "v_rcp_f32 %" #K ", %" #K "\n"
#define PKFMA(K) "v_mov_b32 %" #K ", %" #K "\n"

// packed fp32 fma: 64-bit operands (register pairs); the guide's FP32 vector peak (157.3 TFLOP/s) assumes it issues at the rate of v_fma_f32
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY2(ASM)                                                                                  \
    f2 a0 = {float(threadIdx.x), 1.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;   \
    const f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};                                             \
    for (uint32_t i = 0; i < iters; ++i) {                                                          \
        REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)   \
    }                                                                                               \
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                             \
    if (s.x + s.y == 12345.678f) out[0] = s.x;
#define PK(K) "v_pk_fma_f32 %" #K ", %" #K ", %8, %9\n"
#define PKSEL(K) "v_pk_fma_f32 %" #K ", %" #K ", %8, %8 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
#define PKMUL(K) "v_pk_mul_f32 %" #K ", %" #K ", %8\n"
__global__ __launch_bounds__(256) void kPkFma(uint32_t iters, float* out) { BODY2(PK) }
__global__ __launch_bounds__(256) void kPkFmaSel(uint32_t iters, float* out) { BODY2(PKSEL) }
__global__ __launch_bounds__(256) void kPkMul(uint32_t iters, float* out) { BODY2(PKMUL) }
__global__ __launch_bounds__(256) void kFma(uint32_t iters, float* out) { BODY(FMA) }
__global__ __launch_bounds__(256) void kMax3(uint32_t iters, float* out) { BODY(MAX3) }
__global__ __launch_bounds__(256) void kCvt(uint32_t iters, float* out) { BODY(CVT) }
__global__ __launch_bounds__(256) void kCnd(uint32_t iters, float* out) { BODY(CND) }
__global__ __launch_bounds__(256) void kMul(uint32_t iters, float* out) { BODY(MUL) }
__global__ __launch_bounds__(256) void kCmp(uint32_t iters, float* out) { BODY(CMP) }
__global__ __launch_bounds__(256) void kRcp(uint32_t iters, float* out) { BODY(RCP) }
__global__ __launch_bounds__(256) void kMov(uint32_t iters, float* out) { BODY(PKFMA) }

template <typename K> static int run(const char* name, K k, float* out, uint32_t blocksPerCU) {
    const uint32_t iters = 4096, blocks = 256 * blocksPerCU;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 16u, out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double instPerSimd = double(blocksPerCU) * iters * 64;      // one wave of each block per SIMD, 64 instructions per iteration
    const double cycles = ms * 1e-3 * 2.4e9;
    std::printf("%-18s %u waves/SIMD: %8.3f ms  %6.3f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, blocksPerCU, ms, cycles / instPerSimd);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 64));
    for (uint32_t occ : {1u, 8u}) {
        if (run("v_fma_f32", kFma, out, occ)) return 1;
        if (run("v_pk_fma_f32", kPkFma, out, occ)) return 1;
        if (run("v_pk_fma_f32 op_sel", kPkFmaSel, out, occ)) return 1;
        if (run("v_pk_mul_f32", kPkMul, out, occ)) return 1;
        if (run("v_mul_f32", kMul, out, occ)) return 1;
        if (run("v_max3_f32", kMax3, out, occ)) return 1;
        if (run("v_cvt_f32_ubyte1", kCvt, out, occ)) return 1;
        if (run("v_cndmask_b32", kCnd, out, occ)) return 1;
        if (run("v_cmp_lt_f32", kCmp, out, occ)) return 1;
        if (run("v_mov_b32", kMov, out, occ)) return 1;
        if (run("v_rcp_f32", kRcp, out, occ)) return 1;
    }
    return 0;
}
