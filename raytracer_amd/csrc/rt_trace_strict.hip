// Strict build of the path-tracing kernel: no FMA contraction (this file is compiled with
// -ffp-contract=off), IEEE-rounded sqrt and divide, the reference's rejection samplers with the
// reference's order of random draws.  Also holds the resolve kernel, whose
// arithmetic (chunk sums, 1/spp scale, gamma, quantisation) must round exactly as
// render_tile / linear_to_gamma / to_color4 do (Main.c:975-979, src/base_graphics.c:11-30).
#define RT_STRICT 1
#define RT_NS strict_build
#define RT_KERNEL_FOR trace_kernel_strict
#include "rt_trace_impl.h"

namespace rtmi {

// to_color4(linear_to_gamma(c)) for one channel, src/base_graphics.c:11-30
__device__ __forceinline__ uint8_t to_byte(float c)
{
    float g = 0.0f;
    if (c > 0.0f) g = sqrtf(c);
    const float v = g * 255.0f;
    const float m = (255.0f < v) ? 255.0f : v;
    const float q = (0.0f > m) ? 0.0f : m;
    return (uint8_t)q;
}

__global__ __launch_bounds__(256) void resolve_kernel(const ResolveParams p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.npix) return;
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
    if (p.running) { const float4 r = p.running[i]; tx = r.x; ty = r.y; tz = r.z; }
    for (int c = 0; c < p.nchunks; c++) {
        const float4 v = p.partial[(size_t)c * p.npix + i];
        tx = tx + v.x; ty = ty + v.y; tz = tz + v.z;
    }
    if (p.running) p.running[i] = make_float4(tx, ty, tz, 0.0f);
    const float cx = tx * p.inv_spp, cy = ty * p.inv_spp, cz = tz * p.inv_spp;   // Main.c:977
    if (p.accum) {
        p.accum[3 * (size_t)i + 0] = cx;
        p.accum[3 * (size_t)i + 1] = cy;
        p.accum[3 * (size_t)i + 2] = cz;
    }
    uchar4 px;
    px.x = to_byte(cx); px.y = to_byte(cy); px.z = to_byte(cz); px.w = 255;
    reinterpret_cast<uchar4 *>(p.rgba)[i] = px;
}

hipError_t launch_resolve(const ResolveParams &p, hipStream_t s)
{
    const int threads = 256;
    const int blocks = (int)((p.npix + threads - 1) / threads);
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(threads), 0, s, p);
    return hipGetLastError();
}

}  // namespace rtmi
