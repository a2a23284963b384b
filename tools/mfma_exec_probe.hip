// Does a change of the EXEC mask reach a matrix instruction that has been issued but has not finished?
//
// Each wave issues M back-to-back v_mfma_f32_32x32x16_f16 (8 passes of 4 cycles each), narrows EXEC to lanes 0..47
// right behind them, restores it only when the matrix pipe has drained, and compares every accumulator with the
// same product computed with nothing but waits behind it.  Accumulators are pre-set to a sentinel: a lane whose
// write-back was masked keeps it.  Prints, per variant, the number of wrong accumulator values per lane quarter.
//
// Waits are s_sleep 24 (about 1500 cycles): longer than the matrix instructions of all four waves of a SIMD take.
// Every asm block opens with s_nop 7: the compiler does not see the matrix instruction inside and pads no hazard
// between the vector instructions that produce A and B and the instruction that reads them.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_exec_probe tools/mfma_exec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ f32x16 reference_product(f16x8 a, f16x8 b)
{
    f32x16 r;
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_sleep 24\n\ts_nop 15" : "=&v"(r) : "v"(a), "v"(b));
    return r;
}

// VARIANT 0: EXEC narrowed right behind the matrix instructions (the case in question)
// VARIANT 1: the same instruction stream with EXEC left alone (control)
// VARIANT 2: EXEC narrowed only after a wait that lets the matrix pipe drain (what the product's loop does:
//            its first instruction behind the MFMAs reads an accumulator)
template <int VARIANT, int M>
__global__ __launch_bounds__(1024, 1) void probe(unsigned long long *wrong, int iters)
{
    const int lane = (int)__lane_id();
    unsigned long long bad = 0;
    const unsigned long long narrow = 0x0000ffffffffffffULL;
    for (int it = 0; it < iters; it++) {
        f16x8 a, b;
        for (int k = 0; k < 8; k++) {
            a[k] = (_Float16)(float)(((lane * 7 + k * 3 + it) % 17) - 8);
            b[k] = (_Float16)(float)(((lane * 5 + k * 11 + it * 3) % 13) - 6);
        }
        const f32x16 want = reference_product(a, b);
        f32x16 acc[4];
        for (int m = 0; m < 4; m++)
            for (int i = 0; i < 16; i++) acc[m][i] = -12345.0f;
        if (VARIANT == 0) {
            if (M == 1)
                asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\t"
                             "s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %3\n\ts_sleep 24\n\ts_nop 15\n\ts_mov_b64 exec, s[20:21]\n\t"
                             "s_nop 3"
                             : "+v"(acc[0]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
            else
                asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, 0\n\t"
                             "v_mfma_f32_32x32x16_f16 %2, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, 0\n\t"
                             "s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %6\n\ts_sleep 24\n\ts_nop 15\n\ts_mov_b64 exec, s[20:21]\n\t"
                             "s_nop 3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
        } else if (VARIANT == 3) {
            // EXEC is narrowed, then restored to all lanes M - 1 instructions before the matrix instruction is issued
            // (here M counts the scalar no-ops in between plus one): does the instruction see the restored mask?
            if (M == 1)
                asm volatile("s_nop 7\n\ts_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %3\n\ts_nop 7\n\ts_mov_b64 exec, s[20:21]\n\t"
                             "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_sleep 24\n\ts_nop 15"
                             : "+v"(acc[0]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
            else
                asm volatile("s_nop 7\n\ts_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %3\n\ts_nop 7\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                             "v_nop\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_sleep 24\n\ts_nop 15"
                             : "+v"(acc[0]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
        } else if (VARIANT == 1) {
            asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, 0\n\t"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, 0\n\t"
                         "s_mov_b64 s[20:21], exec\n\ts_nop 0\n\ts_nop 3\n\ts_mov_b64 exec, s[20:21]\n\t"
                         "s_sleep 24\n\ts_nop 15"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
        } else {
            asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, 0\n\t"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, 0\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, 0\n\t"
                         "s_sleep 24\n\ts_nop 15\n\t"
                         "s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %6\n\ts_nop 3\n\ts_mov_b64 exec, s[20:21]\n\ts_nop 3"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b), "s"(narrow) : "s20", "s21");
        }
        for (int m = 0; m < (M == 1 || VARIANT == 3 ? 1 : 4); m++)
            for (int i = 0; i < 16; i++) bad += (acc[m][i] != want[i]) ? 1ULL : 0ULL;
    }
    if (bad) atomicAdd(&wrong[lane >> 4], bad);
}

template <int VARIANT, int M>
static void run(const char *what, unsigned long long *d_wrong, int blocks, int iters)
{
    HIP_OK(hipMemset(d_wrong, 0, 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((probe<VARIANT, M>), dim3(blocks), dim3(1024), 0, 0, d_wrong, iters);
    HIP_OK(hipDeviceSynchronize());
    unsigned long long h[4];
    HIP_OK(hipMemcpy(h, d_wrong, sizeof h, hipMemcpyDeviceToHost));
    const double total = (double)blocks * 1024.0 * iters * 16.0 * (M == 1 || VARIANT == 3 ? 1 : 4);
    printf("%-86s wrong accumulators in lanes 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu of %.3g each\n",
           what, h[0], h[1], h[2], h[3], total / 4.0);
}

int main()
{
    unsigned long long *d_wrong;
    HIP_OK(hipMalloc(&d_wrong, 4 * sizeof(unsigned long long)));
    const int blocks = 256, iters = 2000;            // one 16-wave workgroup per CU: four waves share each SIMD's matrix pipe
    run<1, 4>("4 MFMAs, EXEC untouched (control)", d_wrong, blocks, iters);
    run<2, 4>("4 MFMAs, EXEC narrowed to lanes 0-47 after the matrix pipe has drained", d_wrong, blocks, iters);
    run<0, 1>("1 MFMA, EXEC narrowed to lanes 0-47 right behind it until it has finished", d_wrong, blocks, iters);
    run<0, 4>("4 MFMAs, EXEC narrowed to lanes 0-47 right behind them until they have finished", d_wrong, blocks, iters);
    run<3, 1>("1 MFMA issued right behind the instruction that widens EXEC from lanes 0-47 to all", d_wrong, blocks, iters);
    run<3, 2>("1 MFMA issued one vector instruction behind the s_or that widens EXEC", d_wrong, blocks, iters);
    HIP_OK(hipFree(d_wrong));
    return 0;
}
