// dev probe: does v_fma_mixlo_f16 keep fp16 SUBNORMAL results on gfx950? (v_cvt_f16_f32 does.)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mix_denorm_probe.hip -o scripts/micro/build/mix_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* x, unsigned short* out) {
    const int i = threadIdx.x;
    float v = x[i];
    unsigned a, b;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(v));
    asm volatile("v_mov_b32 %0, 0\n\tv_fma_mixlo_f16 %0, %1, 1.0, 0" : "=&v"(b) : "v"(v));
    out[2 * i] = (unsigned short)a;
    out[2 * i + 1] = (unsigned short)b;
}
int main() {
    float hx[8] = {1e-3f, 6.2e-5f, 3.0e-5f, 1e-5f, 1e-6f, 1e-7f, -2e-6f, 0.5f};
    float* dx; unsigned short* dout; unsigned short ho[16];
    hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(8), 0, 0, dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("x = %10.3e   v_cvt_f16_f32 -> 0x%04x   v_fma_mixlo_f16 -> 0x%04x\n", hx[i], ho[2 * i], ho[2 * i + 1]);
    return 0;
}
