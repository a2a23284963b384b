// Issue-rate table for single VALU instruction forms on gfx950 (16 independent destinations,
// 6 and 8 waves per SIMD), reported in cycles per wave-instruction per SIMD at the shader
// clock measured during the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define BODY(K, STR, ...) if (KIND == K) asm volatile(STR : __VA_ARGS__);

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float x, float y, f2 sx)
{
    f2 p[16];
    float a[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    float vx = x + threadIdx.x * 1e-9f, vy = y + threadIdx.x * 1e-9f;
    f2 xx = f2{vx, vx * 1.5f}, yy = f2{vy, vy * 0.5f};
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            unsigned long long m;
            BODY(0, "v_mul_f32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(1, "v_mul_f32_e32 %0, %1, %0", "+v"(a[i]) : "s"(x))
            BODY(2, "v_add_f32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(3, "v_mul_f32_e32 %0, %1, %2", "=v"(a[i]) : "v"(vx), "v"(vy))
            BODY(4, "v_fma_f32 %0, %1, %2, %0", "+v"(a[i]) : "v"(vx), "v"(vy))
            BODY(5, "v_fmac_f32_e32 %0, %1, %2", "+v"(a[i]) : "v"(vx), "v"(vy))
            BODY(6, "v_cmp_le_f32_e32 vcc, %0, %1", : "v"(vx), "v"(a[i]) : "vcc")
            BODY(7, "v_cmp_le_f32_e32 vcc, %0, %1", : "s"(x), "v"(a[i]) : "vcc")
            BODY(8, "v_cmp_le_f32_e64 %0, %1, %2", "=s"(m) : "s"(x), "v"(a[i]))
            BODY(9, "v_cmp_le_f32_e64 %0, %1, %2", "=s"(m) : "v"(vx), "v"(a[i]))
            BODY(10, "v_pk_fma_f32 %0, %1, %2, %0", "+v"(p[i]) : "v"(xx), "v"(yy))
            BODY(11, "v_pk_fma_f32 %0, %1, %2, %3", "=v"(p[i]) : "v"(xx), "v"(yy), "v"(xx))
            BODY(12, "v_pk_fma_f32 %0, %1, %2, %0", "+v"(p[i]) : "s"(sx), "v"(yy))
            BODY(13, "v_pk_mul_f32 %0, %1, %2", "=v"(p[i]) : "v"(xx), "v"(yy))
            BODY(14, "v_max_f32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(15, "v_and_b32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(16, "v_mov_b32_e32 %0, %1", "=v"(a[i]) : "v"(vx))
            BODY(17, "v_max3_f32 %0, %1, %2, %0", "+v"(a[i]) : "v"(vx), "v"(vy))
            if (KIND == 18) {     // dependent mul then add on one register
                asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(vx));
                asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(vy));
            }
            BODY(19, "v_sub_f32_e32 %0, %1, %0", "+v"(a[i]) : "s"(x))
            BODY(20, "v_dot2c_f32_f16 %0, %1, %2", "+v"(a[i]) : "v"(vx), "v"(vy))
            BODY(21, "v_pk_fma_f16 %0, %1, %2, %0", "+v"(a[i]) : "v"(vx), "v"(vy))
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, int num_cu)
{
    float *out; unsigned long long *clk, h[2];
    CHECK(hipMalloc(&out, sizeof(float) * 256 * num_cu * 8));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f2 sx = {1.0001f, 0.9999f};
    printf("%-44s", name);
    for (int wps = 6; wps <= 8; wps += 2) {
        int blocks = num_cu * wps, iters = 30000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 100, 1.0001f, 0.5f, sx);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f, sx);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double inst = (double)wps * iters * (KIND == 18 ? 32 : 16);
        double mhz = (double)h[0] / (double)h[1] * 100.0;
        printf("  | %d waves: %6.2f ms %5.0f MHz %5.2f cyc", wps, ms, mhz, ms * 1e-3 * mhz * 1e6 / inst);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(clk));
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cu = prop.multiProcessorCount;
    run<0>("v_mul_f32 e32 v,v (acc)", cu);
    run<1>("v_mul_f32 e32 s,v (acc)", cu);
    run<2>("v_add_f32 e32 v,v (acc)", cu);
    run<3>("v_mul_f32 e32 v,v -> other dst", cu);
    run<4>("v_fma_f32 v,v,acc", cu);
    run<5>("v_fmac_f32 e32 v,v", cu);
    run<6>("v_cmp_le_f32 e32 vcc, v, v", cu);
    run<7>("v_cmp_le_f32 e32 vcc, s, v", cu);
    run<8>("v_cmp_le_f32 e64 sdst, s, v", cu);
    run<9>("v_cmp_le_f32 e64 sdst, v, v", cu);
    run<10>("v_pk_fma_f32 v,v,acc", cu);
    run<11>("v_pk_fma_f32 v,v,v -> other dst", cu);
    run<12>("v_pk_fma_f32 s,v,acc", cu);
    run<13>("v_pk_mul_f32 v,v -> other dst", cu);
    run<14>("v_max_f32 e32 v,v", cu);
    run<15>("v_and_b32 e32 v,v", cu);
    run<16>("v_mov_b32 v", cu);
    run<17>("v_max3_f32 v,v,acc", cu);
    run<18>("v_mul_f32 + dependent v_add_f32", cu);
    run<19>("v_sub_f32 e32 s,v", cu);
    run<20>("v_dot2c_f32_f16 v,v", cu);
    run<21>("v_pk_fma_f16 v,v,acc", cu);
    return 0;
}
