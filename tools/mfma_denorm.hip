// Are f16 subnormal inputs of v_mfma_f32_32x32x16_f16 honoured or flushed?  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float aval, float bval) {
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = 0; b[j] = 0; }
    a[0] = (_Float16)aval; b[0] = (_Float16)bval;        // K entry 8*(lane>>5): every lane contributes one product
    f32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = d[0]; out[1] = (float)a[0]; }
}
int main() {
    float *d; hipMalloc(&d, 8); float h[2];
    const float tests[][2] = {{9.5367431640625e-7f, 1024.0f}, {1024.0f, 9.5367431640625e-7f}, {3.0e-5f, 2.0f}, {1.0f, 1.0f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g b=%g: mfma sum over 2 K-halves = %g (expected %g), a as f16 = %g\n", t[0], t[1], h[0], 2.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1], h[1]);
    }
    return 0;
}
