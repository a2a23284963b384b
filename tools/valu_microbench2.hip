// Issue-rate experiments for the sphere-scan loop on gfx950: v_pk_fma_f32 with the operand
// patterns the kernel uses (SGPR pair as src0, broadcast VGPR), v_cmp to an SGPR pair, and the
// kernel's 7 pk_fma + 2 cmp mix.  Also reports the shader clock actually held during each loop
// (s_memtime ticks per s_memrealtime tick; the real-time counter runs at 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float x, float y, f2 sx, f2 sy)
{
    f2 p[16];
    for (int i = 0; i < 16; i++) p[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 1e-3f + i + 0.5f};
    f2 xx = f2{x, x * 1.5f}, yy = f2{y, y * 0.5f};
    unsigned long long cnt = 0;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xx), "v"(yy));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(sx), "v"(yy));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(sx), "v"(yy));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "s"(sx), "v"(yy), "v"(xx));     // no read of dst
            if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "s"(sx), "v"(yy));
            if (KIND == 5) { unsigned long long m; asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "s"(x), "v"(p[i].x)); cnt += m; }
            if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].x) : "s"(x), "v"(y));
            if (KIND == 7) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(p[i].x) : "s"(x));
            if (KIND == 9) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(yy));
        }
        if (KIND == 8) {
            // the scan body twice: 7 pk_fma + 2 v_cmp + s_or per sphere pair
#pragma unroll
            for (int r = 0; r < 2; r++) {
                f2 h, m, e; unsigned long long c0, c1;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(h) : "s"(sx), "v"(p[0]), "v"(p[1]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(m) : "s"(sx), "v"(p[2]), "v"(p[3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(h) : "s"(sy), "v"(p[4]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m) : "s"(sy), "v"(p[5]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(h) : "s"(sx), "v"(p[6]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m) : "s"(sx), "v"(p[7]));
                asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %1, %1, %2" : "=v"(e) : "v"(h), "v"(m));
                asm volatile("s_nop 0\n\tv_cmp_le_f32_e64 %0, %1, %2" : "=s"(c0) : "s"(x), "v"(e.x));
                asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(c1) : "s"(y), "v"(e.y));
                cnt += (c0 | c1) != 0ULL;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = (float)cnt;
    for (int i = 0; i < 16; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, int inst_per_iter, int num_cu)
{
    float *out; unsigned long long *clk, h[2];
    CHECK(hipMalloc(&out, sizeof(float) * 256 * num_cu * 8));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f2 sx = {1.0001f, 0.9999f}, sy = {0.5f, 0.25f};
    for (int wps = 2; wps <= 8; wps += 2) {
        int blocks = num_cu * wps, iters = 40000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 100, 1.0001f, 0.5f, sx, sy);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f, sx, sy);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double inst = (double)wps * iters * inst_per_iter;          // per SIMD
        double mhz = (double)h[0] / (double)h[1] * 100.0;
        printf("%-44s waves/SIMD %d  %7.2f ms  clock %6.0f MHz  %.2f cycles/instr at that clock (%.2f at 2.4 GHz)\n", name, wps, ms, mhz,
               ms * 1e-3 * mhz * 1e6 / inst, ms * 1e-3 * 2.4e9 / inst);
    }
    CHECK(hipFree(out)); CHECK(hipFree(clk));
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    int cu = prop.multiProcessorCount;
    run<0>("v_pk_fma_f32 v,v,v (acc)", 16, cu);
    run<1>("v_pk_fma_f32 s,v,v (acc)", 16, cu);
    run<2>("v_pk_fma_f32 s,v(bcast),v (acc)", 16, cu);
    run<3>("v_pk_fma_f32 s,v,v (dst not read)", 16, cu);
    run<4>("v_pk_mul_f32 s,v", 16, cu);
    run<9>("v_pk_add_f32 v,v", 16, cu);
    run<5>("v_cmp_le_f32_e64 sgpr-dst, s, v  (+2 salu)", 16, cu);
    run<6>("v_fma_f32 s,v,v", 16, cu);
    run<7>("v_mul_f32 s,v", 16, cu);
    run<8>("scan body: 7 pk_fma + 2 v_cmp + salu", 18, cu);
    return 0;
}
