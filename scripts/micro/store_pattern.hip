// dev micro-benchmark: how fast do 256 workgroups push 144 KB each (a 96 x 384 fp32 row block) to HBM, as
//   PATTERN 0  the MFMA accumulator layout of the layer kernels' epilogues: a store instruction = 16 rows x 64 contiguous bytes
//   PATTERN 1  linear: a store instruction = 1 KiB contiguous (what an LDS-staged epilogue would issue)
//   hipcc -O3 --offload-arch=gfx950 -DPATTERN=1 scripts/micro/store_pattern.hip -o scripts/micro/build/store_pattern_1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef PATTERN
#define PATTERN 0
#endif
constexpr int BM = 96, E = 384, THREADS = 512;
__global__ __launch_bounds__(THREADS) void store_kernel(float* __restrict__ out, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rg = wv >> 2, cg = wv & 3, f_row = lane & 15, f_kg = lane >> 4;
    const size_t m0 = (size_t)blockIdx.x * BM;
    f32x4 v = {seed + tid, seed, seed * 2, 1.f};
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        size_t off;
        if (PATTERN == 0) {
            const int rf = i / 6, cf = i % 6;
            const int n = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4;
            off = (m0 + rg * 48 + rf * 16 + f_row) * E + n;
        } else {
            off = m0 * E + (size_t)(i * THREADS + tid) * 4;  // 18 x 512 x 16 B = 144 KB, each wave-instruction 1 KiB contiguous
        }
        *reinterpret_cast<f32x4*>(out + off) = v;
        v[0] += 1.f;
    }
}
int main() {
    float* out;
    const size_t n = (size_t)256 * BM * E;
    hipMalloc(&out, n * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel, dim3(256), dim3(THREADS), 0, 0, out + (size_t)(i % 4) * n, 1.f);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel, dim3(256), dim3(THREADS), 0, 0, out + (size_t)(i % 4) * n, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    printf("PATTERN=%d: %.2f us per launch of 37.7 MB -> %.2f TB/s (incl. launch)\n", PATTERN, best * 1e3, n * 4 / (best * 1e-3) / 1e12);
    return 0;
}
