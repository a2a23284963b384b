// Do matrix and vector instructions of the four waves of a SIMD hide each other, and does it matter where the
// accumulators live?
//
// Every wave runs the shape of the path tracer's tile loop: four v_mfma_f32_32x32x16_f16 (two accumulators, two K
// halves), then the 17 maxima over the 32 accumulators, then V independent v_fma_f32 that stand for the rest of the
// iteration.  Four waves per SIMD (1024-thread workgroups, one per CU), no memory traffic.  Variants:
//   M   products only           V   vector work only (maxima of constants + fillers)
//   MV  both, accumulators in vector registers (what the compiler does with the kernel)
//   MA  both, accumulators in the accumulation registers (asm "a" constraints), read back with v_accvgpr_read_b32
// If the two kinds of work hid each other, time(MV) would be max(time(M), time(V)); if not, their sum.
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_coexec_probe tools/mfma_coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float maxima(const f32x16 &a, const f32x16 &b)
{
    auto one = [](const f32x16 &x) {
        const float g0 = __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), x[2]), g1 = __builtin_fmaxf(__builtin_fmaxf(x[3], x[4]), x[5]);
        const float g2 = __builtin_fmaxf(__builtin_fmaxf(x[6], x[7]), x[8]), g3 = __builtin_fmaxf(__builtin_fmaxf(x[9], x[10]), x[11]);
        const float g4 = __builtin_fmaxf(__builtin_fmaxf(x[12], x[13]), x[14]);
        return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(g0, g1), x[15]), __builtin_fmaxf(__builtin_fmaxf(g2, g3), g4));
    };
    return __builtin_fmaxf(one(a), one(b));
}

// MODE bit 0: products, bit 1: vector work, bit 2: accumulators in AGPRs
template <int MODE, int V>
__global__ __launch_bounds__(1024, 1) void probe(float *out, int iters)
{
    const int lane = (int)__lane_id();
    f16x8 a0, a1, b0, b1, b2, b3;
    for (int k = 0; k < 8; k++) {
        a0[k] = (_Float16)(float)((lane + k) % 5 - 2); a1[k] = (_Float16)(float)((lane * 3 + k) % 7 - 3);
        b0[k] = (_Float16)(float)((lane * 5 + k) % 3 - 1); b1[k] = (_Float16)(float)((lane + 2 * k) % 5 - 2);
        b2[k] = (_Float16)(float)((lane * 7 + k) % 3 - 1); b3[k] = (_Float16)(float)((lane + 3 * k) % 7 - 3);
    }
    float d[8];
    for (int k = 0; k < 8; k++) d[k] = 1.0f + 1e-3f * (float)(lane + k);
    float sink = -1e30f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 c0 = zero, c1 = zero;
    for (int it = 0; it < iters; it++) {
        if (MODE & 1) {
            if (MODE & 4) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %4, 0\n\t"
                             "v_mfma_f32_32x32x16_f16 %1, %2, %5, 0\n\t"
                             "v_mfma_f32_32x32x16_f16 %0, %3, %6, %0\n\t"
                             "v_mfma_f32_32x32x16_f16 %1, %3, %7, %1\n\t"
                             "s_nop 15\n\ts_nop 3"
                             : "=&a"(c0), "=&a"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, zero, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b3, c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 2) {
            if (!(MODE & 1)) asm volatile("" : "+v"(c0), "+v"(c1));        // vector work only: maxima of opaque constants
            float m = maxima(c0, c1);
            asm volatile("" : "+v"(m));
            sink = __builtin_fmaxf(sink, m);
#pragma unroll
            for (int k = 0; k < V / 8; k++)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        } else {
            asm volatile("" :: "v"(c0), "v"(c1));
        }
        // the fragments change a little, so that nothing is hoisted
        asm volatile("" : "+v"(a0), "+v"(a1));
    }
    float s = sink;
    for (int k = 0; k < 8; k++) s += d[k];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int V>
static float run(float *d_out, int iters)
{
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, V>), dim3(256), dim3(1024), 0, 0, d_out, iters / 10);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, V>), dim3(256), dim3(1024), 0, 0, d_out, iters);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int V>
static void row(float *d_out, int iters)
{
    const float m = run<1, V>(d_out, iters), v = run<2, V>(d_out, iters), mv = run<3, V>(d_out, iters), ma = run<7, V>(d_out, iters);
    // cycles per iteration and SIMD at 2.4 GHz nominal: four waves per SIMD run the loop
    const double k = 2.4e6 / (double)iters;
    printf("%3d fillers + 17 maxima per 4 products: products only %.2f ms (%.0f cycles per iteration and wave), vector only %.2f ms (%.0f), "
           "both, accumulators in VGPRs %.2f ms (%.0f; sum %.0f, max %.0f), both, accumulators in AGPRs + 32 v_accvgpr_read %.2f ms (%.0f)\n",
           V, m, m * k / 4, v, v * k / 4, mv, mv * k / 4, (m + v) * k / 4, (m > v ? m : v) * k / 4, ma, ma * k / 4);
}

int main()
{
    float *d_out;
    HIP_OK(hipMalloc(&d_out, 64));
    const int iters = 200000;
    row<0>(d_out, iters);
    row<24>(d_out, iters);
    row<48>(d_out, iters);
    row<72>(d_out, iters);
    row<144>(d_out, iters);
    HIP_OK(hipFree(d_out));
    return 0;
}
