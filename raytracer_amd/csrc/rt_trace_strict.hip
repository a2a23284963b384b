// Strict build of the path-tracing kernel: no FMA contraction (this file is compiled with
// -ffp-contract=off), IEEE-rounded sqrt and divide.  Also holds the resolve kernel, whose
// arithmetic (chunk sums, 1/spp scale, gamma, quantisation) must round exactly as
// render_tile / linear_to_gamma / to_color4 do (Main.c:975-979, src/base_graphics.c:11-30).
#define RT_STRICT 1
#define RT_NS strict_build
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

typedef void (*trace_fn)(const TraceParams);

static trace_fn kernel_for(int mode, int variant)
{
    if (variant == kScanMatrix) return mode ? strict_build::trace_kernel<1, kScanMatrix> : strict_build::trace_kernel<0, kScanMatrix>;
    if (variant == kScanMatrixGlobal) return mode ? strict_build::trace_kernel<1, kScanMatrixGlobal> : strict_build::trace_kernel<0, kScanMatrixGlobal>;
    if (variant == kScanVectorGlobal) return mode ? strict_build::trace_kernel<1, kScanVectorGlobal> : strict_build::trace_kernel<0, kScanVectorGlobal>;
    return mode ? strict_build::trace_kernel<1, kScanVectorLds> : strict_build::trace_kernel<0, kScanVectorLds>;
}

hipError_t prepare_trace_strict(LaunchCache &c, int mode, int variant, size_t lds_bytes)
{
    mode = mode == 1 ? 1 : 0;
    if (c.attr_lds[0][mode][variant] == lds_bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)kernel_for(mode, variant),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) c.attr_lds[0][mode][variant] = lds_bytes;
    return e;
}

hipError_t launch_trace_strict(LaunchCache &c, const TraceParams &p, int variant, int blocks, size_t lds_bytes, hipStream_t s)
{
    const int mode = p.mode == 1 ? 1 : 0;
    hipError_t e = prepare_trace_strict(c, mode, variant, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel_for(mode, variant), dim3(blocks), dim3(block_threads_of(variant)), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu_strict(LaunchCache &c, int variant, size_t lds_bytes)
{
    if (c.occ_lds[0][variant] == lds_bytes) return c.occ_blocks[0][variant];
    int n = 0;
    if (prepare_trace_strict(c, 0, variant, lds_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel_for(0, variant), block_threads_of(variant), lds_bytes) != hipSuccess)
        return 1;
    c.occ_lds[0][variant] = lds_bytes;
    c.occ_blocks[0][variant] = n > 0 ? n : 1;
    return c.occ_blocks[0][variant];
}

}  // namespace rtmi
