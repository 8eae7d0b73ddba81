// dev probe: does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (needed by the split-fp16 operand format, whose
// low halves of small weights are subnormal), and does v_cvt_f16_f32 produce them?  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float av[] = {1.0f, 1e-5f, 3e-6f, 1e-7f, 6e-8f};
    for (float a : av) {
        k<<<1, 64>>>(a, 1.0f, d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%.3e  cvt->f16->f32 %.9e  b=%.3e  mfma sum over K=32: %.9e  (expected %.9e)\n", a, h[1], h[2], h[0], 32.0 * h[1] * h[2]);
    }
    // subnormal x subnormal-free: a subnormal, b large
    k<<<1, 64>>>(2e-6f, 1024.0f, d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("a=2e-6 b=1024: %.9e (expected %.9e)\n", h[0], 32.0 * h[1] * h[2]);
    return 0;
}
