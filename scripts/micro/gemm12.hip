// dev micro-benchmark (round 4): would the twelve-wave form (eight computing waves + four DMA-only waves, pp_ffn_dma.hip) help the dense
// f16x3 Linear layers of ViT-B (today: pp_panel_split.hip, 256 x 192 tiles, eight waves at 250 registers, 50.7 % MFMA busy)?
// Tile 192 x 192, K-step 32 (one 128-byte block per row), three 48 KiB stages, wave tile 48 x 96 (3 x 6 fragments, 54 MFMAs per step).
// Dummy data, no epilogue: timing only.  M = 55 296, K = 768, N = 2304 (the qkv layer at bs 64).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gemm12.hip -o scripts/micro/build/gemm12
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef NSTAGE
#define NSTAGE 3
#endif
constexpr int BM = 192, BN = 192, CW = 8, WAVES = 12, THREADS = WAVES * 64;
constexpr int STAGE = (BM + BN) * 128, LDS = NSTAGE * STAGE;
constexpr int B_OFF = BM * 128;
static_assert(LDS <= 160 * 1024, "LDS");
#define WAITVM_ONLY(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(THREADS) void gemm12_kernel(const char* __restrict__ a, unsigned a_bytes, const char* __restrict__ w, unsigned w_bytes, float* __restrict__ out,
                                                        int K, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 32 workgroups of an XCD = 4 row tiles x 8 column tiles (consecutive ids go round the XCDs)
    const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
    const int per = 4 * ntn;                       // tiles of a band of 4 row tiles
    const int band = (s * 8 + xcd) / per, r = (s * 8 + xcd) % per;
    const int tm = band * 4 + r % 4, tn = r / 4;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ksteps = K / 32;
    if (wv >= CW) {
        const int d = wv - CW;
        const int x_l = lane >> 3;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a), 0, a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, w_bytes, 0x00020000);
        const unsigned va = (unsigned)(m0 + x_l) * (unsigned)(K * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
        const unsigned vw = (unsigned)(n0 + x_l) * (unsigned)(K * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
        auto issue = [&](int k, int st) {
            char* dst = smem + st * STAGE;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + q * 1024), 16, va, k * 128 + q * 8 * K * 4, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + B_OFF + q * 1024), 16, vw, k * 128 + q * 8 * K * 4, 0, 0);
            }
        };
        int st_i = 0;
        for (int k = 0; k < NSTAGE - 1; ++k) { issue(k, st_i); st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1; }
        for (int k = 0; k < ksteps; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            if (NSTAGE == 3) WAITVM_ONLY(12); else WAITVM_ONLY(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue(k + NSTAGE - 1 < ksteps ? k + NSTAGE - 1 : 0, st_i);  // (past the end: a harmless re-read keeps the counts)
            st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
        }
        WAITVM_ONLY(0);
        return;
    }
    const int rg = wv >> 1, cg = wv & 1;
    const int f_row = lane & 15, f_kg = lane >> 4, sw = f_row & 7;
    const int lane_hi = f_row * 128 + ((f_kg ^ sw) << 4), lane_lo = f_row * 128 + (((4 + f_kg) ^ sw) << 4);
    auto opq = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto rd = [&](int lane_off, int uni, int imm) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (lane_off + uni) + imm); };
    f32x4 acc[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int st = 0;
    for (int k = 0; k < ksteps; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int ua = opq(st * STAGE + rg * 48 * 128), ub = opq(st * STAGE + B_OFF + cg * 96 * 128);
        u32x4 ah[3], al[3], bh[6], bl[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) ah[i] = rd(lane_hi, ua, i * 2048);
#pragma unroll
        for (int j = 0; j < 6; ++j) bh[j] = rd(lane_hi, ub, j * 2048);
#pragma unroll
        for (int i = 0; i < 3; ++i) al[i] = rd(lane_lo, ua, i * 2048);
#pragma unroll
        for (int j = 0; j < 6; ++j) bl[j] = rd(lane_lo, ub, j * 2048);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = mma(bh[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = mma(bl[j], ah[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = mma(bh[j], al[i], acc[i][j]);
        st = st + 1 == NSTAGE ? 0 : st + 1;
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) sum += acc[i][j];
    reinterpret_cast<f32x4*>(out)[(size_t)(blockIdx.x % 1024) * THREADS + tid] = sum;
}

int main(int argc, char** argv) {
    const int M = 55296, K = argc > 2 ? atoi(argv[2]) : 768, N = argc > 1 ? atoi(argv[1]) : 2304;
    const size_t abytes = (size_t)M * K * 4, wbytes = (size_t)N * K * 4;
    char *a, *w;
    float* out;
    (void)hipMalloc(&a, abytes);
    (void)hipMalloc(&w, wbytes);
    (void)hipMalloc(&out, 1024 * THREADS * 16);
    std::vector<unsigned short> ha(abytes / 2), hw(wbytes / 2);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    (void)hipMemcpy(a, ha.data(), abytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm12_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int ntn = N / BN, grid = (M / BM) * ntn;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(gemm12_kernel, dim3(grid), dim3(THREADS), LDS, 0, a, (unsigned)abytes, w, (unsigned)wbytes, out, K, ntn);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm12_kernel, dim3(grid), dim3(THREADS), LDS, 0, a, (unsigned)abytes, w, (unsigned)wbytes, out, K, ntn);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms / 5 < best) best = ms / 5;
    }
    const double gf = 2.0 * M * (double)N * K * 1e-9;
    printf("M=%d N=%d K=%d NSTAGE=%d: %.1f us per launch, %.0f TF algorithmic (%.1f %% of 2.5 PF, %.1f %% of the 3-MFMA ceiling), err=%s\n", M, N, K, NSTAGE, best * 1e3,
           gf / best, gf / best / 25.0, gf / best / 8.333, hipGetErrorString(hipGetLastError()));
    return 0;
}
