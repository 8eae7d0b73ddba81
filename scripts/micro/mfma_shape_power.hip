// dev micro-benchmark (round 5): the chip is POWER-limited on the f16x3 path - does the MFMA shape change the energy per FLOP?
// Register-resident operands (no LDS, no memory traffic in the loop), 256 workgroups x 8 waves, N(0,1) fp16 data, and the split-fp16 operand
// mix (hi x hi, lo x hi, hi x lo with lo = fp16(x - hi)): v_mfma_f32_16x16x32_f16 (24 accumulators of 4 registers) against
// v_mfma_f32_32x32x16_f16 (6 accumulators of 16): same FLOPs per launch, same accumulator registers; a 32 x 32 product reads half the
// operand registers per FLOP. Reports TFLOP/s and the shader clock the chip holds (s_memtime against the 100 MHz counter).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_shape_power.hip -o scripts/micro/build/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const u32x4* src, int iters, float* sink, unsigned long long* clk) {
    const int tid = threadIdx.x;
    u32x4 a[6], b[4];
    for (int i = 0; i < 6; ++i) a[i] = src[(i * 512 + tid) % 4096];
    for (int i = 0; i < 4; ++i) b[i] = src[((6 + i) * 512 + tid) % 4096];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][6];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[i]), __builtin_bit_cast(f16x8, a[j]), acc[i][j], 0, 0, 0);
            const u32x4 t = a[0];
#pragma unroll
            for (int j = 0; j < 5; ++j) a[j] = a[j + 1];
            a[5] = t;
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        // 32x32x16: A and B fragments are 8 halves per lane too; 12 MFMAs of twice the FLOPs per pass of the loop
        f32x16 acc[2][3];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[2 * h + i]), __builtin_bit_cast(f16x8, a[3 * h + j]), acc[i][j], 0, 0, 0);
            const u32x4 t = a[0];
#pragma unroll
            for (int j = 0; j < 5; ++j) a[j] = a[j + 1];
            a[5] = t;
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

int main() {
    const int iters = 20000;
    u32x4* src; float* sink; unsigned long long* clk;
    hipMalloc(&src, 4096 * 16); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&clk, 16);
    std::vector<unsigned short> h(4096 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
            float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            if (mode == 1 && ((i / 8 / 512) % 3) == 1) {  // every third fragment holds LOW halves: fp16(x - fp16(x)) of an fp32 value
                float x = g * 1.000123f; _Float16 hi = (_Float16)x; g = x - (float)hi;
            }
            h[i] = f2h(g);
        }
        hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int shape : {16, 32}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f; unsigned long long c[2] = {0, 0};
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, src, iters, sink, clk);
                else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, src, iters, sink, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost); }
            }
            const double flop = 256.0 * 8 * iters * 24 * 2.0 * 16 * 16 * 32;
            printf("%-28s %dx%d: %8.1f us  %6.0f TFLOP/s  shader clock %4.0f MHz\n", mode ? "split mix (1/3 low halves)" : "N(0,1) fp16", shape, shape, best * 1e3,
                   flop / (best * 1e-3) * 1e-12, (double)c[0] / ((double)c[1] / 100.0));
        }
    }
    return 0;
}
