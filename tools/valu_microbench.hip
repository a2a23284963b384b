// Measures the vector-FP32 issue rates that bound the sphere-test loop on gfx950:
// v_fma_f32, v_pk_fma_f32, v_mul_f32+v_add_f32, each with 16 independent accumulators,
// at 1, 2, 4 and 8 waves per SIMD.  Prints TFLOP/s (fma = 2 flop/lane, pk_fma = 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float x, float y)
{
    float a[16];
    float2v p[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = float2v{a[i], a[i] + 0.5f}; }
    float2v xx = float2v{x, x * 1.5f}, yy = float2v{y, y * 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xx), "v"(yy));
            if (KIND == 2) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                             asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y)); }
            if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(x), "v"(y));
            if (KIND == 4) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(xx));
                             asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(yy)); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, double flop_per_lane_iter, int num_cu)
{
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * num_cu * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = num_cu * wps;       // 256 threads = 4 waves = 1 wave per SIMD per block
        int iters = 20000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double flop = (double)blocks * 256 * iters * flop_per_lane_iter;
        double inst = (double)blocks * 4 * iters * ((KIND == 2 || KIND == 4) ? 32 : 16);
        printf("%-28s waves/SIMD %d  %.2f ms  %.1f TFLOP/s  (%.2f cycles/wave-instr/SIMD at 2.4 GHz)\n", name, wps, ms,
               flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (inst / ((double)num_cu * 4)));
    }
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    int cu = prop.multiProcessorCount;
    run<0>("v_fma_f32", 32, cu);
    run<3>("v_fma_f32 (sgpr operand)", 32, cu);
    run<1>("v_pk_fma_f32", 64, cu);
    run<2>("v_mul_f32 + v_add_f32", 32, cu);
    run<4>("v_pk_mul_f32 + v_pk_add_f32", 64, cu);
    return 0;
}
