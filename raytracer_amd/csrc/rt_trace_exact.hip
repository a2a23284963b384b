// Exact build of the path-tracing kernel, the default: no FMA contraction (this file is compiled with
// -ffp-contract=off), IEEE-rounded sqrt and divide, every operation of the path in the reference's
// order, and loop-free samplers (RT_DIRECT) in place of the reference's rejection loops.  Bit-identical
// to the CPU evaluation of the same expressions; in deterministic mode bit-identical to the reference.
#define RT_STRICT 1
#define RT_DIRECT 1
#define RT_NS exact_build
#define RT_KERNEL_FOR trace_kernel_exact
#include "rt_trace_impl.h"
