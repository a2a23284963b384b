#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t pk(float a, float b) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 v = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(uint32_t, v); }
__global__ void k(const float *x, uint32_t *out)
{
    float a = x[threadIdx.x], b = x[threadIdx.x + 64];
    uint32_t r0 = pk(a, a);
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(r0) : "v"(a));
    uint32_t r2 = pk(b, b);
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r2) : "v"(b));
    out[threadIdx.x] = r0; out[threadIdx.x + 64] = r2;
}
int main()
{
    float h[128]; uint32_t o[128];
    for (int i = 0; i < 128; i++) h[i] = (i - 60) * 1.2345678f + 1e-3f * i * i;
    float *d; uint32_t *od; hipMalloc(&d, sizeof h); hipMalloc(&od, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, od); hipMemcpy(o, od, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; i++) {
        _Float16 x1 = (_Float16)h[i]; _Float16 x2 = (_Float16)(h[i] - (float)x1);
        uint16_t b1, b2; memcpy(&b1, &x1, 2); memcpy(&b2, &x2, 2);
        uint32_t want = i < 64 ? (b1 | ((uint32_t)b2 << 16)) : (b2 | ((uint32_t)b1 << 16));
        if (o[i] != want) { bad++; if (bad < 5) printf("i %d got %08x want %08x\n", i, o[i], want); }
    }
    printf("mix split: %d mismatches of 128\n", bad);
    return bad != 0;
}
