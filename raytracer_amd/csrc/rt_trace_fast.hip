// Fast build of the path-tracing kernel: FMA contraction, unit-length ray directions,
// hardware rsqrt / rcp / sqrt.  Same estimator as the strict build; images agree
// statistically (PSNR-tested), not bit for bit.
#define RT_STRICT 0
#define RT_NS fast_build
#include "rt_trace_impl.h"

namespace rtmi {

// The dynamic-LDS attribute and the occupancy answer depend only on (kernel, LDS bytes): both
// are looked up once per change of scene size, not once per frame.
static size_t g_attr_lds[2] = {(size_t)-1, (size_t)-1};

static hipError_t prepare(int mode, size_t lds_bytes)
{
    if (g_attr_lds[mode] == lds_bytes) return hipSuccess;
    const void *k = mode ? (const void *)fast_build::trace_kernel<1> : (const void *)fast_build::trace_kernel<0>;
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) g_attr_lds[mode] = lds_bytes;
    return e;
}

hipError_t launch_trace_fast(const TraceParams &p, int blocks, size_t lds_bytes, hipStream_t s)
{
    hipError_t e = prepare(p.mode == 1 ? 1 : 0, lds_bytes);
    if (e != hipSuccess) return e;
    auto k = (p.mode == 1) ? fast_build::trace_kernel<1> : fast_build::trace_kernel<0>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(kBlockThreads), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu_fast(size_t lds_bytes)
{
    static size_t cached_lds = (size_t)-1;
    static int cached_n = 1;
    if (cached_lds == lds_bytes) return cached_n;
    int n = 0;
    if (prepare(0, lds_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fast_build::trace_kernel<0>, kBlockThreads, lds_bytes) != hipSuccess)
        return 1;
    cached_lds = lds_bytes;
    cached_n = n > 0 ? n : 1;
    return cached_n;
}

void trace_forget_fast() { g_attr_lds[0] = g_attr_lds[1] = (size_t)-1; }

}  // namespace rtmi
