// FAST arithmetic: reciprocal direction hoisted out of the slab test, fused multiply-add allowed.
#include "lrl_device.h"
#define LRL_FAST 1
#define LRL_NS lrl_fast
#include "lrl_kernels.inc"
#include "lrl_lean.inc"

hipError_t lrl_launch_render_fast(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters, bool pool,
                                  hipStream_t stream) {
    uint32_t tx = (fr.width + LRL_TILE - 1) / LRL_TILE, ty = (fr.rowEnd - fr.rowBegin + LRL_TILE - 1) / LRL_TILE;
    dim3 grid(tx * ty), block(LRL_BLOCK_THREADS);
    size_t lds = size_t(fr.stackEntries) * LRL_BLOCK_THREADS * 8;
    using namespace lrl_fast;
    if (counters) {
        if (pool) hipLaunchKernelGGL((renderKernel<true, true>), grid, block, lds, stream, sc, fr, out, hit, counters);
        else hipLaunchKernelGGL((renderKernel<false, true>), grid, block, lds, stream, sc, fr, out, hit, counters);
    } else {
        if (pool) hipLaunchKernelGGL((renderKernel<true, false>), grid, block, lds, stream, sc, fr, out, hit, counters);
        else hipLaunchKernelGGL((renderKernel<false, false>), grid, block, lds, stream, sc, fr, out, hit, counters);
    }
    return hipGetLastError();
}
