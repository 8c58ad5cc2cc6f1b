// PARITY arithmetic: compiled with -ffp-contract=off; expression order as on llvmpipe.
// Also holds the two streaming pre-compute kernels, whose outputs are compared bit for bit.
#include "lrl_device.h"
#define LRL_FAST 0
#define LRL_NS lrl_parity
#include "lrl_kernels.inc"

namespace lrl_parity {

// TrianglePre, /root/reference/BVH/shaders/TrianglePreCompute0.cpp:55-80.  One lane per triangle:
// reads the 160-B origin record (three 16-B + uv loads), writes the 64-B hot and 32-B cold records.
// The hot record carries the triangle's number in the y of its last quarter: the four-wide kernels read a copy of the hot records in the
// order of their tree's leaves (triLeaf[posOfTri[i]], lrl_build.hip) and report hits by that number; the copy is written here too when
// the scene has one (triLeaf != NULL; posOfTri[i] = LRL_REF_NONE: the tree holds no such triangle).
__global__ __launch_bounds__(256) void trianglePreKernel(const LrlTriangle* __restrict__ in, const uint32_t* __restrict__ material,
                                                         uint32_t first, uint32_t count, float4* __restrict__ triHot,
                                                         LrlTriCold* __restrict__ triCold, float4* __restrict__ triLeaf, const uint32_t* __restrict__ posOfTri) {
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
    const float4 h0 = make_float4(n.x, n.y, n.z, -dot3(n, p1)), h1 = make_float4(p1.x, p1.y, p1.z, k1.x), h2 = make_float4(k1.y, k1.z, k2.x, k2.y),
                 h3 = make_float4(k2.z, __uint_as_float(first + i), __uint_as_float(material[i]), 0.0f);
    h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
    if (triLeaf) {
        const uint32_t p = posOfTri[first + i];
        if (p != LRL_REF_NONE) { float4* l = triLeaf + size_t(p) * 4; l[0] = h0; l[1] = h1; l[2] = h2; l[3] = h3; }
    }
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

// ---- Pool water simulation: /root/reference/Pool/shaders/WaterAccelerationCompute0.cpp:22-41,
// WaterPositionCompute0.cpp:49-55, WaterPositionSyncCompute0.cpp:49-75, driven as Pool/Pool.cpp:224-235 ----
// The reference dispatches 2 * n + 1 tiny grids per frame (n = 200: 401 launches for 2304 points).  Here ONE workgroup keeps
// the heights in LDS and runs all n substeps between barriers, then scatters the heights into the corners of the water
// triangles (device copy of their origin records); TrianglePre of those triangles follows as a second launch.
// Same expressions in the same order, no contraction (this translation unit): bit-identical to the shaders on llvmpipe.
#define LRL_WATER_THREADS 1024u
#define LRL_WATER_MAX_POINTS 16384u          /* 64 KB of LDS */
__global__ __launch_bounds__(LRL_WATER_THREADS) void waterRunKernel(float4* __restrict__ water, uint32_t gx, uint32_t gy, float k, float dt,
                                                                  uint32_t substeps, LrlTriangle* __restrict__ tris) {
    extern __shared__ float zs[];
    const uint32_t n = gx * gy;
    for (uint32_t i = threadIdx.x; i < n; i += LRL_WATER_THREADS) zs[i] = water[i].x;
    __syncthreads();
    // every thread owns the points threadIdx.x + j * 1024; their v and a stay in registers
    constexpr uint32_t OWN = LRL_WATER_MAX_POINTS / LRL_WATER_THREADS;
    float v[OWN], acc[OWN];
#pragma unroll
    for (uint32_t j = 0; j < OWN; ++j) {
        const uint32_t i = threadIdx.x + j * LRL_WATER_THREADS;
        v[j] = i < n ? water[i].y : 0.0f;
        acc[j] = i < n ? water[i].z : 0.0f;
    }
    for (uint32_t s = 0; s < substeps; ++s) {
#pragma unroll
        for (uint32_t j = 0; j < OWN; ++j) {
            const uint32_t index = threadIdx.x + j * LRL_WATER_THREADS;
            if (index < n) {
                const uint32_t x = index % gx, y = index / gx;
                float dv = 0;
                if (x != 0 && x != gx - 1) dv += 2 * zs[index] - zs[index - 1] - zs[index + 1];
                else if (x != 0) dv += zs[index] - zs[index - 1];
                else dv += zs[index] - zs[index + 1];
                const uint32_t offset = gy;                      // the shader's row stride for the y neighbours (:33)
                if (y != 0 && y != offset - 1) dv += 2 * zs[index] - zs[index - offset] - zs[index + offset];
                else if (y != 0) dv += zs[index] - zs[index - offset];
                else dv += zs[index] - zs[index + offset];
                acc[j] = k * dt * dv;
            }
        }
        __syncthreads();                                         // every acceleration of this substep has read its neighbours
#pragma unroll
        for (uint32_t j = 0; j < OWN; ++j) {
            const uint32_t index = threadIdx.x + j * LRL_WATER_THREADS;
            if (index < n) {
                zs[index] -= dt * (v[j] + acc[j] / 2);
                v[j] += acc[j];
                v[j] *= 0.999985f;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t j = 0; j < OWN; ++j) {
        const uint32_t index = threadIdx.x + j * LRL_WATER_THREADS;
        if (index < n) {
            const float z = zs[index];
            water[index] = make_float4(z, v[j], acc[j], water[index].w);
            if (tris) {                                          // WaterPositionSync: each corner is written by exactly one point
                const uint32_t x = index % gx, y = index / gx;
                const uint32_t t = x + y * 2 * (gx - 1);
                if (x != 0) {
                    if (y != 0) tris[t - gx].p1[2] = z;
                    if (y != gy - 1) { tris[t - 1].p2[2] = z; tris[t + gx - 2].p3[2] = z; }
                }
                if (x != gx - 1) {
                    if (y != 0) { tris[t - 2 * (gx - 1)].p3[2] = z; tris[t - gx + 1].p2[2] = z; }
                    if (y != gy - 1) tris[t].p1[2] = z;
                }
            }
        }
    }
}

}  // namespace lrl_parity

namespace lrl_parity {
// float -> unorm8 as OpenGL converts a fragment colour for an RGBA8 framebuffer: clamp to [0, 1], scale by 255, round to nearest
__global__ __launch_bounds__(256) void quantizeKernel(const float4* __restrict__ in, uchar4* __restrict__ out, unsigned long long pixels) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= pixels) return;
    const float4 v = in[i];
    auto q = [](float x) { return (unsigned char)rintf(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f); };      // NaN -> 0 (fmaxf returns the other operand)
    out[i] = make_uchar4(q(v.x), q(v.y), q(v.z), q(v.w));
}
}  // namespace lrl_parity

hipError_t lrl_launch_quantize(const float4* in, uchar4* out, unsigned long long pixels, hipStream_t stream) {
    if (!pixels) return hipSuccess;
    hipLaunchKernelGGL(lrl_parity::quantizeKernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, in, out, pixels);
    return hipGetLastError();
}

hipError_t lrl_launch_water_run(float4* water, uint32_t gx, uint32_t gy, float k, float dt, uint32_t substeps, LrlTriangle* tris, hipStream_t stream) {
    hipLaunchKernelGGL(lrl_parity::waterRunKernel, dim3(1), dim3(LRL_WATER_THREADS), size_t(gx) * gy * 4, stream, water, gx, gy, k, dt, substeps, tris);
    return hipGetLastError();
}

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
                                   LrlTriCold* triCold, float4* triLeaf, const uint32_t* posOfTri, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_parity::trianglePreKernel, dim3((count + 255) / 256), dim3(256), 0, stream, in, material, first, count, triHot, triCold,
                       posOfTri ? triLeaf : nullptr, posOfTri);
    return hipGetLastError();
}

hipError_t lrl_launch_circle_pre(LrlCircle* circles, uint32_t count, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_parity::circlePreKernel, dim3((count + 255) / 256), dim3(256), 0, stream, circles, count);
    return hipGetLastError();
}
