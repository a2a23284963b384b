// One ray-sphere discriminant as the 31-term f16 inner product of the matrix scan, computed by two chained
// v_mfma_f32_32x32x16_f16 and compared with exact arithmetic (also with the K entries reversed and per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A[32 rows][32 k], B[32 k][32 cols] as f16; out = full 32x32 of the two chained instructions
__global__ void k(const _Float16 *A, const _Float16 *B, float *out)
{
    const int l = threadIdx.x;
    f32x16 d = {0};
    for (int s = 0; s < 2; s++) {
        f16x8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = A[(l & 31) * 32 + 16 * s + 8 * (l >> 5) + j]; b[j] = B[(16 * s + 8 * (l >> 5) + j) * 32 + (l & 31)]; }
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[r];
}
int main()
{
    // the 31 K entries of the worst calibration case: sphere side, ray side
    float o[3] = {-8.93609715f, 0.294345111f, -0.982466638f}, u[3] = {-0.813530028f, 0.318556488f, -0.486508667f};
    double c[3] = {-49.8822746, -48.8722496, -1.01292324}, r = 4.0262537;
    float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2], g = u[0] * o[0] + u[1] * o[1] + u[2] * o[2];
    float rf[10] = {u[0] * u[0], u[1] * u[1], u[2] * u[2], 2 * u[0] * u[1], 2 * u[0] * u[2], 2 * u[1] * u[2], 2 * (o[0] - g * u[0]), 2 * (o[1] - g * u[1]), 2 * (o[2] - g * u[2]), g * g - oo};
    float sf[10] = {(float)(c[0] * c[0]), (float)(c[1] * c[1]), (float)(c[2] * c[2]), (float)(c[0] * c[1]), (float)(c[0] * c[2]), (float)(c[1] * c[2]), (float)c[0], (float)c[1], (float)c[2],
                    (float)(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] - r * r)};
    _Float16 se[32] = {0}, re[32] = {0};
    for (int f = 0; f < 9; f++) {
        _Float16 x1 = (_Float16)rf[f], x2 = (_Float16)(rf[f] - (float)x1), c1 = (_Float16)sf[f], c2 = (_Float16)(sf[f] - (float)c1);
        se[3 * f] = c1; se[3 * f + 1] = c1; se[3 * f + 2] = c2; re[3 * f] = x1; re[3 * f + 1] = x2; re[3 * f + 2] = x1;
    }
    { _Float16 x1 = (_Float16)rf[9], x2 = (_Float16)(rf[9] - (float)x1), c1 = (_Float16)sf[9], c2 = (_Float16)(sf[9] - (float)c1);
      se[27] = 1; se[28] = 1; re[27] = x1; re[28] = x2; se[29] = c1; se[30] = c2; re[29] = -1; re[30] = -1; }
    double exact = 0; for (int e = 0; e < 32; e++) exact += (double)se[e] * (double)re[e];
    std::vector<_Float16> A(32 * 32, 0), B(32 * 32, 0);
    // row 0 / col 0: the case; row 1 / col 1: same with the order of the K entries reversed; row 2 / col 2: only entries 0..15; row 3/col 3: only 16..31
    for (int e = 0; e < 32; e++) { A[0 * 32 + e] = se[e]; B[e * 32 + 0] = re[e]; A[1 * 32 + e] = se[31 - e]; B[e * 32 + 1] = re[31 - e]; }
    for (int e = 0; e < 16; e++) { A[2 * 32 + e] = se[e]; B[e * 32 + 2] = re[e]; A[3 * 32 + 16 + e] = se[16 + e]; B[(16 + e) * 32 + 3] = re[16 + e]; }
    double e0 = 0, e1 = 0; for (int e = 0; e < 16; e++) { e0 += (double)se[e] * (double)re[e]; e1 += (double)se[16 + e] * (double)re[16 + e]; }
    _Float16 *dA, *dB; float *dO;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dO, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    float out[1024]; hipMemcpy(out, dO, 4096, hipMemcpyDeviceToHost);
    printf("exact %.6f  hw row0col0 %.6f  reversed %.6f  first16 hw %.6f exact %.6f  last16 hw %.6f exact %.6f\n", exact, out[0], out[33], out[2 * 32 + 2], e0, out[3 * 32 + 3], e1);
    printf("cross: row0 col1 %.6f row1 col0 %.6f\n", out[1], out[32]);
    return 0;
}
