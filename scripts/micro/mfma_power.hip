// Microbenchmark (dev only): what the MFMA pipes sustain on REGISTER-RESIDENT operands - no LDS, no memory traffic in
// the loop - as a function of the operand data, and the shader clock the chip holds meanwhile (s_memtime cycles
// against the 100 MHz s_memrealtime). 256 workgroups x 8 waves (two per SIMD), each wave 24 independent accumulators,
// 4 + 6 operand fragments loaded once. Data: zeros, constant, N(0,1) bf16 (what the convolutions see). The clock is taken
// over wave 0's loop only: the SIMD arbiter favours the older of its two waves (wave 0 gets ~3/4 of the MFMA slots and is
// done after 60 % of the launch), so the TFLOP/s figure comes from the launch time, the MHz figure from that window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(const u32x4* src, int iters, float* sink, unsigned long long* clk, unsigned long long* span) {
    const int tid = threadIdx.x;
    u32x4 a[6], b[4];
    for (int i = 0; i < 6; ++i) a[i] = src[(i * 512 + tid) % 4096];
    for (int i = 0; i < 4; ++i) b[i] = src[((6 + i) * 512 + tid) % 4096];
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[i]), __builtin_bit_cast(bf16x8, a[j]), acc[i][j], 0, 0, 0);
        // rotate the fragments so that consecutive MFMAs do not see the same operand registers forever
        const u32x4 t = a[0];
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] = a[j + 1];
        a[5] = t;
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sink[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    if (blockIdx.x == 0 && tid == 0) { span[0] = r0; span[1] = r1; }
}

// the same MFMA stream with its operands re-read from LDS at the convolution kernels' ratio (10 ds_read_b128 per 24 MFMAs,
// register double-buffered), still no global traffic in the loop: what the fragment reads cost in clock
__global__ __launch_bounds__(512, 2) void k_lds(const u32x4* src, int iters, float* sink, unsigned long long* clk, unsigned long long* span) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 8192; i += 512) reinterpret_cast<u32x4*>(smem)[i] = src[i % 4096];  // 128 KiB
    __syncthreads();
    const int f_row = lane & 15, f_kg = lane >> 4;
    const char* base = smem + (wv >> 2) * 12288 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
    const char* wbase = smem + 65536 + (wv & 3) * 8192 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a[2][6], b[2][4];
    for (int j = 0; j < 6; ++j) a[0][j] = *reinterpret_cast<const u32x4*>(base + j * 2048);
    for (int i = 0; i < 4; ++i) b[0][i] = *reinterpret_cast<const u32x4*>(wbase + i * 2048);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = ((it + s + 1) & 3) * 64;  // the other K half / stage: addresses change every step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[s ^ 1][i] = *reinterpret_cast<const u32x4*>(wbase + ((i * 2048 + off) ^ ((it & 4) << 10)));
#pragma unroll
            for (int j = 0; j < 6; ++j) a[s ^ 1][j] = *reinterpret_cast<const u32x4*>(base + ((j * 2048 + off) ^ ((it & 4) << 12)));
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[s][i]), __builtin_bit_cast(bf16x8, a[s][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (q < 9) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float sm = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) sm += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sink[blockIdx.x * 512 + tid] = sm;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; span[0] = r0; span[1] = r1; }
}

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int n16 = 4096 * 8;  // bf16 elements
    std::vector<uint16_t> h(n16);
    u32x4* d; float* sink; unsigned long long* clk;
    hipMalloc(&d, n16 * 2); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&clk, 16 + 256 * 32); unsigned long long* span; hipMalloc(&span, 16 * 16);
    const int iters = argc > 1 ? atoi(argv[1]) : 6000; const int grid = argc > 2 ? atoi(argv[2]) : 256; const int only = argc > 3 ? atoi(argv[3]) : -1;  // 24 MFMAs x 6000 x 16 cycles ~ 1.2 ms at 2 waves / SIMD
    const char* names[] = {"zeros", "constant 1.0", "N(0,1) bf16", "N(0,1) x N(0,1/sqrt K) (activations x weights)", "uniform bits"};
    for (int mode = 0; mode < 5; ++mode) {
        if (only >= 0 && mode != only) continue;
        srand(1);
        for (int i = 0; i < n16; ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            float v = mode == 0 ? 0.f : mode == 1 ? 1.f : mode == 2 ? g : mode == 3 ? (((i / 8 / 512) >= 6) ? g * 0.017f : g) : 0.f;
            h[i] = mode == 4 ? (uint16_t)(rand() & 0x7f7f) : bf16_of(v);  // (mode 4: random mantissa / exponent bits, finite)
        }
        hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, sink, clk, span);
        hipEventRecord(e0);
        const int reps = 10;
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, sink, clk, span + 2 * w);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        const double flop = (double)grid * 8 * 24 * iters * 16384.0;
        printf("%-48s %8.1f us  %7.0f TFLOP/s  shader clock %4.0f MHz\n", names[mode], ms * 1e3, flop / ms / 1e9, (double)c[0] / c[1] * 100.0);
    }
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int mode = 0; mode < 3; mode += 2) {
        srand(1);
        for (int i = 0; i < n16; ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            h[i] = mode == 0 ? 0 : bf16_of(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
        }
        hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), 131072, 0, d, iters, sink, clk, span);
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), 131072, 0, d, iters, sink, clk, span);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        printf("LDS-fed (10 ds_read_b128 per 24 MFMAs), %-12s %8.1f us  %7.0f TFLOP/s  shader clock %4.0f MHz\n", mode == 0 ? "zeros" : "N(0,1) bf16", ms * 1e3,
               (double)grid * 8 * 24 * iters * 16384.0 / ms / 1e9, (double)c[0] / c[1] * 100.0);
    }
    return 0;
}
