// Fast build of the path-tracing kernel: FMA contraction, unit-length ray directions,
// hardware rsqrt / rcp / sqrt.  Same estimator as the strict build; images agree
// statistically (PSNR-tested), not bit for bit.
#define RT_STRICT 0
#define RT_NS fast_build
#include "rt_trace_impl.h"

namespace rtmi {

hipError_t launch_trace_fast(const TraceParams &p, int blocks, size_t lds_bytes, hipStream_t s)
{
    auto k = (p.mode == 1) ? fast_build::trace_kernel<1> : fast_build::trace_kernel<0>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(kBlockThreads), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu_fast(size_t lds_bytes)
{
    int n = 0;
    hipFuncSetAttribute((const void *)fast_build::trace_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fast_build::trace_kernel<0>, kBlockThreads, lds_bytes) != hipSuccess) return 1;
    return n > 0 ? n : 1;
}

}  // namespace rtmi
