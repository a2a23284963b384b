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
    unsigned long long mask = 0x5555555555555555ULL ^ (unsigned long long)iters;
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
            BODY(22, "v_sub_co_u32_e32 %0, vcc, %1, %0", "+v"(a[i]) : "v"(vx) : "vcc")
            BODY(23, "v_sub_co_u32_e64 %0, %1, %2, %3", "=v"(a[i]), "=s"(m) : "v"(vx), "v"(vy))
            BODY(24, "v_cndmask_b32_e32 %0, %1, %0, vcc", "+v"(a[i]) : "v"(vx) : "vcc")
            BODY(25, "v_mul_lo_u32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(26, "v_rsq_f32_e32 %0, %0", "+v"(a[i]))
            BODY(27, "v_sqrt_f32_e32 %0, %0", "+v"(a[i]))
            BODY(28, "v_sin_f32_e32 %0, %0", "+v"(a[i]))
            BODY(29, "v_lshl_add_u32 %0, %1, 8, %0", "+v"(a[i]) : "v"(vx))
            BODY(30, "v_xor_b32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(31, "v_lshrrev_b32_e32 %0, 15, %0", "+v"(a[i]))
            BODY(32, "v_cvt_f32_u32_e32 %0, %0", "+v"(a[i]))
            BODY(33, "v_or3_b32 %0, %1, %2, %0", "+v"(a[i]) : "v"(vx), "v"(vy))
            BODY(34, "v_fma_f32 %0, %1, %2, %0 clamp", "+v"(a[i]) : "v"(vx), "v"(vy))
            BODY(35, "v_pk_fma_f32 %0, %1, %2, %0 clamp", "+v"(p[i]) : "v"(xx), "v"(yy))
            BODY(36, "v_add_u32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
            BODY(38, "v_cndmask_b32_e64 %0, %1, %0, %2", "+v"(a[i]) : "v"(vx), "s"(mask))
            if (KIND == 39) {     // compare, then select on its result
                asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc" : "+v"(a[i]) : "v"(vx), "v"(vy) : "vcc");
            }
            if (KIND == 40) {     // compare into an SGPR pair, then select on it
                asm volatile("v_cmp_lt_f32_e64 %1, %2, %0\n\tv_cndmask_b32_e64 %0, %3, %0, %1" : "+v"(a[i]), "=&s"(m) : "v"(vx), "v"(vy));
            }
            BODY(37, "v_min_f32_e32 %0, %1, %0", "+v"(a[i]) : "v"(vx))
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
        double inst = (double)wps * iters * ((KIND == 18 || KIND == 39 || KIND == 40) ? 32 : 16);
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
    run<22>("v_sub_co_u32 e32 vcc", cu);
    run<23>("v_sub_co_u32 e64 sdst", cu);
    run<24>("v_cndmask_b32 e32 (vcc)", cu);
    run<25>("v_mul_lo_u32", cu);
    run<26>("v_rsq_f32", cu);
    run<27>("v_sqrt_f32", cu);
    run<28>("v_sin_f32", cu);
    run<29>("v_lshl_add_u32", cu);
    run<30>("v_xor_b32 e32", cu);
    run<31>("v_lshrrev_b32 e32 imm", cu);
    run<32>("v_cvt_f32_u32", cu);
    run<33>("v_or3_b32", cu);
    run<34>("v_fma_f32 clamp", cu);
    run<35>("v_pk_fma_f32 clamp", cu);
    run<36>("v_add_u32 e32", cu);
    run<37>("v_min_f32 e32", cu);
    run<38>("v_cndmask_b32 e64 (sgpr pair mask)", cu);
    run<39>("v_cmp e32 vcc + v_cndmask e32 (per instr)", cu);
    run<40>("v_cmp e64 sdst + v_cndmask e64 (per instr)", cu);
    return 0;
}
