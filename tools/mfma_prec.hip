// How v_mfma_f32_32x32x16_f16 sums: single products, cancelling pairs in the same / another group of eight K
// entries, a large term next to a tiny one.  hipcc --offload-arch=gfx950 -O2 -o mfma_prec mfma_prec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// product 0: a0*b0 ; product 1: a1*b1 ; rest zero.  One output element.
__global__ void k(float *out, float a0, float b0, float a1, float b1, int pos1) {
    const int lane = threadIdx.x, h = lane >> 5;
    h8 a, b;
    for (int j = 0; j < 8; j++) { int kk = 8 * h + j; float x = 0, y = 0; if (!(lane & 31)) { if (kk == 0) { x = a0; y = b0; } if (kk == pos1) { x = a1; y = b1; } } a[j] = (_Float16)x; b[j] = (_Float16)y; }
    f32x16 c; for (int i = 0; i < 16; i++) c[i] = 0;
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = d[0];
}
int main() {
    float *d; hipMalloc(&d, 4); float r;
    const float u = 1.0f + ldexpf(1.0f, -10);
    struct { float a0, b0, a1, b1; int pos; const char *what; } cs[] = {
        {u, u, 0, 0, 1, "single product (1+2^-10)^2"},
        {123.0f * u, 97.0f * u, 0, 0, 1, "single product 123u*97u"},
        {u, u, -1.0f, 1.0f, 1, "(1+2^-10)^2 - 1, same group"},
        {u, u, -1.0f, 1.0f, 8, "(1+2^-10)^2 - 1, other group"},
        {128.0f * u, 64.0f * u, -128.0f, 64.0f, 1, "8192 u^2 - 8192 same group"},
        {128.0f * u, 64.0f * u, -128.0f, 64.0f, 8, "8192 u^2 - 8192 other group"},
        {128.0f, 64.0f, 0.01f * u, 0.013f * u, 1, "8192 + tiny product, same group"},
        {128.0f, 64.0f, 0.01f * u, 0.013f * u, 8, "8192 + tiny product, other group"},
    };
    for (auto &c : cs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.a0, c.b0, c.a1, c.b1, c.pos);
        hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
        double exact = (double)(float)(_Float16)c.a0 * (double)(float)(_Float16)c.b0 + (double)(float)(_Float16)c.a1 * (double)(float)(_Float16)c.b1;
        printf("%-36s: got %.10g exact %.10g  err %.3g (f32-rounded exact %.10g)\n", c.what, r, exact, r - exact, (double)(float)exact);
    }
    return 0;
}
