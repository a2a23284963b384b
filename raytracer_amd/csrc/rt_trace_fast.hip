// Fast build of the path-tracing kernel: FMA contraction, unit-length ray directions,
// hardware rsqrt / rcp / sqrt.  Same estimator as the strict build; images agree
// statistically (PSNR-tested), not bit for bit.
#define RT_STRICT 0
#define RT_NS fast_build
#include "rt_trace_impl.h"

namespace rtmi {

typedef void (*trace_fn)(const TraceParams);

static trace_fn kernel_for(int mode, int variant)
{
    if (variant == kScanMatrix) return mode ? fast_build::trace_kernel<1, kScanMatrix> : fast_build::trace_kernel<0, kScanMatrix>;
    if (variant == kScanMatrixGlobal) return mode ? fast_build::trace_kernel<1, kScanMatrixGlobal> : fast_build::trace_kernel<0, kScanMatrixGlobal>;
    if (variant == kScanVectorGlobal) return mode ? fast_build::trace_kernel<1, kScanVectorGlobal> : fast_build::trace_kernel<0, kScanVectorGlobal>;
    return mode ? fast_build::trace_kernel<1, kScanVectorLds> : fast_build::trace_kernel<0, kScanVectorLds>;
}

hipError_t prepare_trace_fast(LaunchCache &c, int mode, int variant, size_t lds_bytes)
{
    mode = mode == 1 ? 1 : 0;
    if (c.attr_lds[1][mode][variant] == lds_bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)kernel_for(mode, variant),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) c.attr_lds[1][mode][variant] = lds_bytes;
    return e;
}

hipError_t launch_trace_fast(LaunchCache &c, const TraceParams &p, int variant, int blocks, size_t lds_bytes, hipStream_t s)
{
    const int mode = p.mode == 1 ? 1 : 0;
    hipError_t e = prepare_trace_fast(c, mode, variant, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel_for(mode, variant), dim3(blocks), dim3(block_threads_of(variant)), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu_fast(LaunchCache &c, int variant, size_t lds_bytes)
{
    if (c.occ_lds[1][variant] == lds_bytes) return c.occ_blocks[1][variant];
    int n = 0;
    if (prepare_trace_fast(c, 0, variant, lds_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel_for(0, variant), block_threads_of(variant), lds_bytes) != hipSuccess)
        return 1;
    c.occ_lds[1][variant] = lds_bytes;
    c.occ_blocks[1][variant] = n > 0 ? n : 1;
    return c.occ_blocks[1][variant];
}

}  // namespace rtmi
