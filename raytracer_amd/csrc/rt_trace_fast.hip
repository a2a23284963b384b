// Fast build of the path-tracing kernel: FMA contraction, unit-length ray directions,
// hardware rsqrt / rcp / sqrt / sin / cos.  Same estimator as the other builds; images agree
// statistically (PSNR-tested), not bit for bit.
#define RT_STRICT 0
#define RT_NS fast_build
#define RT_KERNEL_FOR trace_kernel_fast
#include "rt_trace_impl.h"
