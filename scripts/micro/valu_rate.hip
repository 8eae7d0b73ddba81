// dev only: issue rate of single VALU instructions on one wave of gfx950 (cycles per wave64 instruction), 16 independent
// destinations per loop trip so that no result is waited for. hipcc --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL(NAME, ASM)                                                                         \
    __global__ void NAME(float* out, long long* clk, int n) {                                     \
        float r[16];                                                                              \
        float a = out[threadIdx.x], b = 1.0001f;                                                  \
        for (int i = 0; i < 16; ++i) r[i] = a + i;                                                \
        long long t0 = __builtin_amdgcn_s_memtime(); long long r0 = __builtin_amdgcn_s_memrealtime();                                              \
        for (int it = 0; it < n; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b)); \
        }                                                                                         \
        asm volatile("s_nop 0" ::: "memory");                                                     \
        long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();                                              \
        float s = 0;                                                                              \
        for (int i = 0; i < 16; ++i) s += r[i];                                                   \
        out[threadIdx.x + blockIdx.x * blockDim.x] = s;                                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMax((unsigned long long*)&clk[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&clk[1], (unsigned long long)(r1 - r0)); }                               \
    }
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_exp32, "v_exp_f32 %0, %0")
KERNEL(k_exp16, "v_exp_f16 %0, %0")
KERNEL(k_log32, "v_log_f32 %0, %0")
KERNEL(k_rcp32, "v_rcp_f32 %0, %0")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_cvtpk, "v_cvt_pk_bf16_f32 %0, %0, %1")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %0")
KERNEL(k_cvt_f32, "v_cvt_f32_f16 %0, %0")
KERNEL(k_cvtpk_f16, "v_cvt_pk_f16_f32 %0, %0, %1")
KERNEL(k_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,0,0]")
KERNEL(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,0]")
// packed fp32: 64-bit register pairs
#define KERNEL2(NAME, ASM)                                                                        \
    __global__ void NAME(float* out, long long* clk, int n) {                                     \
        typedef float f2 __attribute__((ext_vector_type(2)));                                     \
        f2 r[16];                                                                                 \
        float a0 = out[threadIdx.x];                                                              \
        f2 a = {a0, a0 + 1.f}, b = {1.0001f, 0.9999f};                                            \
        for (int i = 0; i < 16; ++i) r[i] = a + (float)i;                                         \
        long long t0 = __builtin_amdgcn_s_memtime(); long long r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < n; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b)); \
        }                                                                                         \
        asm volatile("s_nop 0" ::: "memory");                                                     \
        long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime(); \
        float s = 0;                                                                              \
        for (int i = 0; i < 16; ++i) s += r[i][0] + r[i][1];                                      \
        out[threadIdx.x + blockIdx.x * blockDim.x] = s;                                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMax((unsigned long long*)&clk[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&clk[1], (unsigned long long)(r1 - r0)); } \
    }
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL2(k_pkmul, "v_pk_mul_f32 %0, %0, %2")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %2")
template <typename K>
void run(const char* name, K k, int waves) {
    float* out; long long* clk;
    hipMalloc(&out, 1 << 20); hipMalloc(&clk, 16);
    hipMemset(out, 0, 1 << 20);
    const int n = 200000;
    k<<<1, 64 * waves>>>(out, clk, n);
    hipMemset(clk, 0, 16);
    k<<<1, 64 * waves>>>(out, clk, n);
    hipDeviceSynchronize();
    long long cc[2]; hipMemcpy(cc, clk, 16, hipMemcpyDeviceToHost); long long c = cc[0];
    // s_memtime counts at a fixed 100 MHz on gfx950? report raw ticks per instruction; compare against v_fma (4 cycles)
    printf("%-10s waves/CU %d: %.3f memtime ticks, %.3f ns (memrealtime, 100 MHz) per instruction per wave (slowest wave)\n", name, waves, (double)c / (n * 16.0), (double)cc[1] * 10.0 / (n * 16.0));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int w : {1, 4, 8, 12}) {
        run("v_pk_fma_f32", k_pkfma, w); run("v_pk_mul_f32", k_pkmul, w); run("v_pk_add_f32", k_pkadd, w); run("v_mul_f32", k_mul, w);
        run("v_cvt_f16_f32", k_cvt_f16, w); run("v_cvt_f32_f16", k_cvt_f32, w); run("v_cvt_pk_f16_f32", k_cvtpk_f16, w);
        run("v_fma_mix_f32", k_mix, w); run("v_fma_mixlo_f16", k_mixlo, w);
        run("v_fma_f32", k_fma, w); run("v_add_f32", k_add, w); run("v_exp_f32", k_exp32, w); run("v_exp_f16", k_exp16, w);
        run("v_log_f32", k_log32, w); run("v_rcp_f32", k_rcp32, w); run("v_sqrt_f32", k_sqrt, w); run("v_max3_f32", k_max3, w);
        run("v_cvt_pk", k_cvtpk, w); run("v_ldexp", k_ldexp, w);
    }
    return 0;
}
