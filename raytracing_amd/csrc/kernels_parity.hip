// PARITY arithmetic: compiled with -ffp-contract=off; expression order as on llvmpipe.
// Also holds the two streaming pre-compute kernels, whose outputs are compared bit for bit.
#include "lrl_device.h"
#define LRL_FAST 0
#define LRL_NS lrl_parity
#include "lrl_kernels.inc"

namespace lrl_parity {

// TrianglePre, /root/reference/BVH/shaders/TrianglePreCompute0.cpp:55-80.  One lane per triangle:
// reads the 160-B origin record (three 16-B + uv loads), writes the 64-B hot and 32-B cold records.
__global__ __launch_bounds__(256) void trianglePreKernel(const LrlTriangle* __restrict__ in, const uint32_t* __restrict__ material,
                                                         uint32_t first, uint32_t count, float4* __restrict__ triHot,
                                                         LrlTriCold* __restrict__ triCold) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4* q = reinterpret_cast<const float4*>(in + i);
    float4 a = q[0], b = q[1], c = q[2], uvA = q[3], uvB = q[4];
    f3 p1 = xyz(a);
    f3 e1 = xyz(b) - p1, e2 = xyz(c) - p1;
    f3 n = cross3(e1, e2);
    float s = dot3(n, n);
    n = normalize3(n);
    float d = dot3(e1, e2);
    f3 k1 = (dot3(e2, e2) * e1 - d * e2) / s;
    f3 k2 = (dot3(e1, e1) * e2 - d * e1) / s;
    float4* h = triHot + size_t(first + i) * 4;
    h[0] = make_float4(n.x, n.y, n.z, -dot3(n, p1));
    h[1] = make_float4(p1.x, p1.y, p1.z, k1.x);
    h[2] = make_float4(k1.y, k1.z, k2.x, k2.y);
    h[3] = make_float4(k2.z, 0.0f, 0.0f, 0.0f);
    float4* cold = reinterpret_cast<float4*>(triCold + first + i);
    cold[0] = uvA;                                                      // uv1, uv2
    cold[1] = make_float4(uvB.x, uvB.y, __uint_as_float(material[i]), 0.0f);   // uv3, material
}

// CirclePre, /root/reference/BVH/shaders/CirclePreCompute0.cpp:35-44 (in place)
__global__ __launch_bounds__(256) void circlePreKernel(LrlCircle* __restrict__ circles, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float4* q = reinterpret_cast<float4*>(circles + i);
    float4 plane = q[0], sph = q[1], e = q[2];
    f3 n = normalize3(xyz(plane));
    q[0] = make_float4(n.x, n.y, n.z, -dot3(n, xyz(sph)));
    f3 e1 = normalize3(xyz(e));
    q[2] = make_float4(e1.x, e1.y, e1.z, e.w);
}

}  // namespace lrl_parity

static size_t ldsBytes(const KFrame& fr) { return size_t(fr.stackEntries) * LRL_BLOCK_THREADS * 8; }
static dim3 gridFor(const KFrame& fr) {
    uint32_t tx = (fr.width + LRL_TILE - 1) / LRL_TILE, ty = (fr.rowEnd - fr.rowBegin + LRL_TILE - 1) / LRL_TILE;
    return dim3(tx * ty);
}

hipError_t lrl_launch_render_parity(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters, bool pool,
                                    hipStream_t stream) {
    dim3 grid = gridFor(fr), block(LRL_BLOCK_THREADS);
    size_t lds = ldsBytes(fr);
    using namespace lrl_parity;
    if (counters) {
        if (pool) hipLaunchKernelGGL((renderKernel<true, true>), grid, block, lds, stream, sc, fr, out, hit, counters);
        else hipLaunchKernelGGL((renderKernel<false, true>), grid, block, lds, stream, sc, fr, out, hit, counters);
    } else {
        if (pool) hipLaunchKernelGGL((renderKernel<true, false>), grid, block, lds, stream, sc, fr, out, hit, counters);
        else hipLaunchKernelGGL((renderKernel<false, false>), grid, block, lds, stream, sc, fr, out, hit, counters);
    }
    return hipGetLastError();
}

hipError_t lrl_launch_decay_origin(const KScene& sc, const LrlTrans* trans, uint32_t stackEntries, bool pool, LrlDecayOrigin* out,
                                   hipStream_t stream) {
    size_t lds = size_t(stackEntries) * LRL_BLOCK_THREADS * 8;
    using namespace lrl_parity;
    if (pool) hipLaunchKernelGGL((decayOriginKernel<true>), dim3(1), dim3(LRL_BLOCK_THREADS), lds, stream, sc, trans, stackEntries, out);
    else hipLaunchKernelGGL((decayOriginKernel<false>), dim3(1), dim3(LRL_BLOCK_THREADS), lds, stream, sc, trans, stackEntries, out);
    return hipGetLastError();
}

hipError_t lrl_launch_triangle_pre(const LrlTriangle* in, const uint32_t* material, uint32_t first, uint32_t count, float4* triHot,
                                   LrlTriCold* triCold, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_parity::trianglePreKernel, dim3((count + 255) / 256), dim3(256), 0, stream, in, material, first, count, triHot, triCold);
    return hipGetLastError();
}

hipError_t lrl_launch_circle_pre(LrlCircle* circles, uint32_t count, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_parity::circlePreKernel, dim3((count + 255) / 256), dim3(256), 0, stream, circles, count);
    return hipGetLastError();
}
