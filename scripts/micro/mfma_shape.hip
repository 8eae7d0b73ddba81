// dev micro-benchmark (the one bf16 experiment asked for): the bf16 MFMA stream of a 64 x 96 wave tile with its
// operands re-read from LDS every K step (register double-buffered, no global traffic in the loop), tiled two ways:
//   SHAPE 16: v_mfma_f32_16x16x32_bf16 - per K = 32: 4 + 6 fragments (ds_read_b128 each), 24 MFMAs
//   SHAPE 32: v_mfma_f32_32x32x16_bf16 - per K = 16: 2 + 3 fragments, 6 MFMAs  (= per K = 32: 10 fragments, 12 MFMAs)
// Same accumulator registers (96), same LDS bytes per FLOP: a fragment is 16 bytes per lane either way, and the bytes a wave
// tile needs per K step are (rows + columns) x K x 2 whatever instruction multiplies them. 256 workgroups x 8 waves.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_shape.hip -o scripts/micro/build/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const u32x4* src, int ksteps32, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 8192; i += 512) reinterpret_cast<u32x4*>(smem)[i] = src[i % 4096];  // 128 KiB
    __syncthreads();
    float s = 0.f;
    if (SHAPE == 16) {
        const int f_row = lane & 15, f_kg = lane >> 4;
        const char* abase = smem + (wv >> 2) * 12288 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
        const char* bbase = smem + 65536 + (wv & 3) * 8192 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
        f32x4 acc[4][6];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 a[2][6], b[2][4];
        for (int j = 0; j < 6; ++j) a[0][j] = *reinterpret_cast<const u32x4*>(abase + j * 2048);
        for (int i = 0; i < 4; ++i) b[0][i] = *reinterpret_cast<const u32x4*>(bbase + i * 2048);
        for (int it = 0; it < ksteps32; it += 2) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int off = ((it + st + 1) & 3) * 64;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) b[st ^ 1][i] = *reinterpret_cast<const u32x4*>(bbase + ((i * 2048 + off) ^ ((it & 4) << 10)));
#pragma unroll
                for (int j = 0; j < 6; ++j) a[st ^ 1][j] = *reinterpret_cast<const u32x4*>(abase + ((j * 2048 + off) ^ ((it & 4) << 12)));
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[st][i]), __builtin_bit_cast(bf16x8, a[st][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#pragma unroll
                for (int q = 0; q < 10; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (q < 9) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        const int f_row = lane & 31, f_kg = lane >> 5;  // 32 rows x (2 groups of 8 k): 16 bytes per lane
        const char* abase = smem + (wv >> 2) * 12288 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
        const char* bbase = smem + 65536 + (wv & 3) * 8192 + f_row * 128 + ((f_kg ^ (f_row & 7)) << 4);
        f32x16 acc[2][3];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        u32x4 a[2][3], b[2][2];
        for (int j = 0; j < 3; ++j) a[0][j] = *reinterpret_cast<const u32x4*>(abase + j * 4096);
        for (int i = 0; i < 2; ++i) b[0][i] = *reinterpret_cast<const u32x4*>(bbase + i * 4096);
        for (int it = 0; it < 2 * ksteps32; it += 2) {  // K = 16 per step
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int off = ((it + st + 1) & 3) * 32;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) b[st ^ 1][i] = *reinterpret_cast<const u32x4*>(bbase + ((i * 4096 + off) ^ ((it & 4) << 10)));
#pragma unroll
                for (int j = 0; j < 3; ++j) a[st ^ 1][j] = *reinterpret_cast<const u32x4*>(abase + ((j * 4096 + off) ^ ((it & 4) << 12)));
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[st][i]), __builtin_bit_cast(bf16x8, a[st][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    }
    sink[blockIdx.x * 512 + tid] = s;
}

int main() {
    std::vector<uint16_t> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff)) | (uint16_t)((rand() & 1) << 15);  // bf16 of magnitude ~1
    u32x4* d;
    float* sink;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&sink, 256 * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int ksteps = 6000;
    const double flop = 256.0 * 8 * ksteps * 2.0 * 64 * 96 * 32;
    for (int shape : {16, 32, 16, 32}) {
        auto kern = shape == 16 ? k<16> : k<32>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d, ksteps, sink);
        hipEventRecord(e0);
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d, ksteps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%s: %.3f ms per launch, %.0f TFLOP/s (err %s)\n", shape == 16 ? "16x16x32 bf16 (10 reads / 24 MFMAs)" : "32x32x16 bf16 (10 reads / 12 MFMAs)", ms,
               flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
