// How soon after a vector instruction writes a register may a matrix instruction read it as an operand?
//
// Each wave corrupts one dword of its A fragment, repairs it with a v_mov_b32 inside an asm block and issues
// v_mfma_f32_32x32x16_f16 on the fragment K instructions later (K = 0, 1, 2, 4 v_nop in between).  A product that
// differs from the reference was computed from the corrupted dword: the matrix instruction read the register
// before the write had landed.  The compiler pads no hazards inside an asm block, so the block shows the bare
// hardware behaviour; compiled code is padded according to the compiler's own hazard table.
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_valu_raw_probe tools/mfma_valu_raw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(unsigned long long *wrong, int iters)
{
    const int lane = (int)__lane_id();
    unsigned long long bad = 0;
    for (int it = 0; it < iters; it++) {
        f16x8 a, b;
        for (int k = 0; k < 8; k++) {
            a[k] = (_Float16)(float)(((lane * 7 + k * 3 + it) % 17) - 8);
            b[k] = (_Float16)(float)(((lane * 5 + k * 11 + it * 3) % 13) - 6);
        }
        f32x16 want;
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_sleep 24\n\ts_nop 15" : "=&v"(want) : "v"(a), "v"(b));
        const u32x4 frag = __builtin_bit_cast(u32x4, a);
        const unsigned other = 0x48004800u;    // (8, 8): not the value the product needs
        f32x16 got;
        // the fragment is assembled in v[100:103] with dword 3 wrong, then dword 3 is repaired K instructions before the product
        asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\ts_nop 7\n\t"
                     "v_mov_b32 v103, %5\n\t"
                     ".rept %c6\n\tv_nop\n\t.endr\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, v[100:103], %7, 0\n\ts_sleep 24\n\ts_nop 15"
                     : "=&v"(got) : "v"(frag[0]), "v"(frag[1]), "v"(frag[2]), "v"(other), "v"(frag[3]), "n"(K), "v"(b)
                     : "v100", "v101", "v102", "v103");
        for (int i = 0; i < 16; i++) bad += (got[i] != want[i]) ? 1ULL : 0ULL;
    }
    if (bad) atomicAdd(&wrong[0], bad);
}

template <int K, int WAVES>
static void run(unsigned long long *d_wrong, int blocks, int iters)
{
    HIP_OK(hipMemset(d_wrong, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL((probe<K, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d_wrong, iters);
    HIP_OK(hipDeviceSynchronize());
    unsigned long long h;
    HIP_OK(hipMemcpy(&h, d_wrong, sizeof h, hipMemcpyDeviceToHost));
    printf("%d waves per CU, %d v_nop between the v_mov_b32 and the v_mfma that reads its result: %llu of %.3g accumulators wrong\n",
           WAVES, K, h, (double)blocks * WAVES * 64.0 * iters * 16.0);
}

int main()
{
    unsigned long long *d_wrong;
    HIP_OK(hipMalloc(&d_wrong, sizeof(unsigned long long)));
    const int blocks = 256, iters = 2000;
    run<0, 16>(d_wrong, blocks, iters); run<1, 16>(d_wrong, blocks, iters); run<2, 16>(d_wrong, blocks, iters); run<4, 16>(d_wrong, blocks, iters);
    run<0, 4>(d_wrong, blocks, iters); run<1, 4>(d_wrong, blocks, iters); run<2, 4>(d_wrong, blocks, iters); run<4, 4>(d_wrong, blocks, iters);
    HIP_OK(hipFree(d_wrong));
    return 0;
}
