// Fast build of the path-tracing kernel: FMA contraction, unit-length ray directions,
// hardware rsqrt / rcp / sqrt.  Same estimator as the strict build; images agree
// statistically (PSNR-tested), not bit for bit.
#define RT_STRICT 0
#define RT_NS fast_build
#include "rt_trace_impl.h"

namespace rtmi {

// The dynamic-LDS attribute and the occupancy answer depend only on (kernel, LDS bytes): both
// are looked up once per change of scene size, not once per frame.
typedef void (*trace_fn)(const TraceParams);

static trace_fn kernel_for(int mode, bool geom_global)
{
    if (geom_global) return mode ? fast_build::trace_kernel<1, true> : fast_build::trace_kernel<0, true>;
    return mode ? fast_build::trace_kernel<1, false> : fast_build::trace_kernel<0, false>;
}

static size_t g_attr_lds[2][2] = {{(size_t)-1, (size_t)-1}, {(size_t)-1, (size_t)-1}};

static hipError_t prepare(int mode, bool geom_global, size_t lds_bytes)
{
    if (g_attr_lds[mode][geom_global] == lds_bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)kernel_for(mode, geom_global),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) g_attr_lds[mode][geom_global] = lds_bytes;
    return e;
}

hipError_t launch_trace_fast(const TraceParams &p, bool geom_global, int blocks, size_t lds_bytes, hipStream_t s)
{
    const int mode = p.mode == 1 ? 1 : 0;
    hipError_t e = prepare(mode, geom_global, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel_for(mode, geom_global), dim3(blocks), dim3(kBlockThreads), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu_fast(bool geom_global, size_t lds_bytes)
{
    static size_t cached_lds[2] = {(size_t)-1, (size_t)-1};
    static int cached_n[2] = {1, 1};
    if (cached_lds[geom_global] == lds_bytes) return cached_n[geom_global];
    int n = 0;
    if (prepare(0, geom_global, lds_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel_for(0, geom_global), kBlockThreads, lds_bytes) != hipSuccess)
        return 1;
    cached_lds[geom_global] = lds_bytes;
    cached_n[geom_global] = n > 0 ? n : 1;
    return cached_n[geom_global];
}

void trace_forget_fast()
{
    for (int m = 0; m < 2; m++) for (int g = 0; g < 2; g++) g_attr_lds[m][g] = (size_t)-1;
}

}  // namespace rtmi
