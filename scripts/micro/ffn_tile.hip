// dev micro-benchmark: the second GEMM of the fused layer kernel in isolation (96 LDS-resident rows x a stream of W2
// tiles through the ring of eight 8 KiB LDS-DMA slots), with the workgroup organised two ways:
//   WAVES = 8: 2 x 4 waves, wave tile 48 rows x 96 columns (what pp_mlp.hip does: 18 MFMAs and 9 fragment reads a step)
//   WAVES = 4: 1 x 4 waves with up to 512 registers, wave tile 96 x 96 (36 MFMAs and 12 fragment reads a step)
// Both run the same number of MFMAs per SIMD; the question is what the halved LDS fragment traffic per FLOP buys.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize scripts/micro/ffn_tile.hip -o scripts/micro/build/ffn_tile
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 96, E = 384, F = 1536, CHUNK = 128;
constexpr int ROW_BYTES = 128, HS_KB = BM * ROW_BYTES, SLOT = 8192;
#ifndef NSLOT_
#define NSLOT_ 8
#endif
constexpr int NSLOT = NSLOT_;  // 8 (what the layer kernel can afford) or 12 (to see what a deeper ring would buy)
constexpr int OFF_GS = 0, OFF_RING = 2 * HS_KB, LDS = OFF_RING + NSLOT * SLOT;
constexpr unsigned OOB = 0x7ffffff0u;
#ifndef ABL  // timing-only ablations: 1 no DMA, 2 no fragment reads, 4 no barrier, 8 no MFMA
#define ABL 0
#endif
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int WAVES>
__device__ __forceinline__ void ffn_tile_body(const __bf16* __restrict__ W2, unsigned w2_bytes, float* __restrict__ out, int nsteps8) {
    constexpr int RF = WAVES == 8 ? 3 : 6;   // row fragments per wave
    constexpr int IPS = 8 / WAVES;            // DMA instructions per slot and wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = WAVES == 8 ? wv >> 2 : 0, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W2), 0, w2_bytes, 0x00020000);
    char* const ring = smem + OFF_RING;

    // the G rows: any deterministic bf16 pattern (both organisations see the same bytes)
    for (int i = tid; i < 2 * HS_KB / 4; i += WAVES * 64)
        reinterpret_cast<unsigned*>(smem + OFF_GS)[i] = 0x3c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    __syncthreads();

    const unsigned row_bytes = F * 2u;
    const int d_lc = (lane & 7) ^ (lane >> 3);
    unsigned lane_off[IPS];
#pragma unroll
    for (int u = 0; u < IPS; ++u) {
        const int d_line = (wv * IPS + u) * 8 + (lane >> 3);
        lane_off[u] = (unsigned)(d_line + 192 * (d_lc >> 2)) * row_bytes + (unsigned)((d_lc & 3) << 4);
    }
    // slot q (= 3 j + third) of step s: chunk c = (s / 4) % 12, k-step j = s % 4
    auto issue_slot = [&](int s, int third, int pos) {
        const int c = (s >> 2) % 12, j = s & 3;
        const unsigned base = (unsigned)(third * 64) * row_bytes + (unsigned)((c * CHUNK + 32 * j) * 2);
        if (ABL & 1) return;
        if (ABL & 16) {  // timing only: weights pre-packed in consumption order, every slot 8 KiB contiguous
#pragma unroll
            for (int u = 0; u < IPS; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (lds_ptr_t)(ring + pos * SLOT + (wv * IPS + u) * 1024), 16,
                                                         (unsigned)(((c * 4 + j) * 3 + third) * SLOT + (wv * IPS + u) * 1024 + lane * 16), 0, 0, 0);
            return;
        }
#pragma unroll
        for (int u = 0; u < IPS; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (lds_ptr_t)(ring + pos * SLOT + (wv * IPS + u) * 1024), 16, base + lane_off[u], 0, 0, 0);
    };
    const int frag_sw = f_row & 7;
    const int rows0 = rg * 48 + f_row;
    auto read_w = [&](int pos0, u32x4 (&wf)[6]) {
        if (ABL & 2) { for (int nf = 0; nf < 6; ++nf) asm volatile("" : "=v"(wf[nf])); return; }
        const int ch = (((cg >> 1) * 4 + f_kg) ^ frag_sw) << 4;
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) {
            const int line0 = (cg & 1) * 96 + nf * 16;
            const int pos = (pos0 + (line0 >> 6)) % NSLOT;
            wf[nf] = *reinterpret_cast<const u32x4*>(ring + pos * SLOT + ((line0 & 63) + f_row) * ROW_BYTES + ch);
        }
    };
    auto read_rows = [&](int j, u32x4 (&gf)[RF]) {
        if (ABL & 2) { for (int rf = 0; rf < RF; ++rf) asm volatile("" : "=v"(gf[rf])); return; }
        const char* gbase = smem + OFF_GS + (j >> 1) * HS_KB + rows0 * ROW_BYTES + ((((j & 1) * 4 + f_kg) ^ frag_sw) << 4);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) gf[rf] = *reinterpret_cast<const u32x4*>(gbase + rf * 16 * ROW_BYTES);
    };

    f32x4 acc[RF][6];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) acc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wb[2][6], gb[2][RF];

    // prologue: slots 0..7 in flight, the first step's fragments in registers
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) issue_slot(q / 3, q % 3, q);
    wait_dma_and_barrier<(NSLOT - 3) * IPS>();
    read_w(0, wb[0]);
    read_rows(0, gb[0]);

    for (int it = 0; it < nsteps8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // eight steps = 24 slots: the ring positions repeat
            const int s = it * 8 + u, cur = u & 1;
            wait_dma_and_barrier<(NSLOT - 6) * IPS>();  // slots of step s + 1 have landed; everybody has step s in registers
#pragma unroll
            for (int i = 0; i < 3; ++i) {  // refill this step's slots with stream index + 8
                const int q = 3 * u + i + NSLOT;  // stream index relative to the start of the iteration
                issue_slot(it * 8 + q / 3, q % 3, (3 * u + i) % NSLOT);
            }
            read_w((3 * (u + 1)) % NSLOT, wb[cur ^ 1]);
            read_rows((s + 1) & 3, gb[cur ^ 1]);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) { if (!(ABL & 8)) acc[rf][nf] = mma(wb[cur][nf], gb[cur][rf], acc[rf][nf]); }
            if (WAVES == 8) {
#pragma unroll
                for (int i = 0; i < 9; ++i) { SGB(0x008, 2); SGB(0x100, 1); if (i == 0 || i == 3 || i == 6) SGB(0x010, 1); }
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) { SGB(0x008, 3); SGB(0x100, 1); if (i % 2 == 0) SGB(0x010, 1); }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
    float sum = 0.f;
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) sum += acc[rf][nf][0] + acc[rf][nf][1] + acc[rf][nf][2] + acc[rf][nf][3];
    atomicAdd(out + blockIdx.x, sum);
}

// (plain kernels around the template body: the host pass rejected the kernel template itself without a diagnostic)
__global__ __launch_bounds__(512, 2) void ffn_tile_kernel8(const __bf16* W2, unsigned w2_bytes, float* out, int nsteps8) {
    ffn_tile_body<8>(W2, w2_bytes, out, nsteps8);
}
__global__ __launch_bounds__(256, 1) void ffn_tile_kernel4(const __bf16* W2, unsigned w2_bytes, float* out, int nsteps8) {
    ffn_tile_body<4>(W2, w2_bytes, out, nsteps8);
}

// WAVES = 8 again, but every step is [192 outputs x 64 k] instead of [384 outputs x 32 k]: the same 24 KiB and 18 MFMAs per
// wave, fetched as whole 128-byte row segments (the L2 -> LDS stream moves 21 TB/s with those, 13 TB/s with 64-byte ones).
// A wave owns 48 columns in each half of the outputs; steps alternate between the halves.
__global__ __launch_bounds__(512, 2) void ffn_tile_kernel8_k64(const __bf16* W2, unsigned w2_bytes, float* out, int nsteps8) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W2), 0, w2_bytes, 0x00020000);
    char* const ring = smem + OFF_RING;
    for (int i = tid; i < 2 * HS_KB / 4; i += 512)
        reinterpret_cast<unsigned*>(smem + OFF_GS)[i] = 0x3c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    __syncthreads();
    const unsigned row_bytes = F * 2u;
    const int d_lc = (lane & 7) ^ (lane >> 3);
    const unsigned lane_off = (unsigned)(wv * 8 + (lane >> 3)) * row_bytes + (unsigned)(d_lc << 4);
    // step s: chunk c = (s / 4) % 12, k-block kb = (s >> 1) & 1, half = s & 1
    auto issue_slot = [&](int s, int third, int pos) {
        const int c = (s >> 2) % 12, kb = (s >> 1) & 1, half = s & 1;
        const unsigned base = (unsigned)(half * 192 + third * 64) * row_bytes + (unsigned)((c * CHUNK + 64 * kb) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (lds_ptr_t)(ring + pos * SLOT + wv * 1024), 16, base + lane_off, 0, 0, 0);
    };
    const int frag_sw = f_row & 7;
    const int rows0 = rg * 48 + f_row;
    auto read_frags = [&](int pos0, int kb, u32x4 (&wf)[3][2], u32x4 (&gf)[3][2]) {
        const char* gbase = smem + OFF_GS + kb * HS_KB + rows0 * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + f_kg) ^ frag_sw) << 4;
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                const int line0 = cg * 48 + nf * 16;
                const int pos = (pos0 + (line0 >> 6)) % NSLOT;
                wf[nf][ks] = *reinterpret_cast<const u32x4*>(ring + pos * SLOT + ((line0 & 63) + f_row) * ROW_BYTES + ch);
            }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) gf[rf][ks] = *reinterpret_cast<const u32x4*>(gbase + rf * 16 * ROW_BYTES + ch);
        }
    };
    f32x4 acc[3][6];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) acc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wb[2][3][2], gb[2][3][2];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) issue_slot(q / 3, q % 3, q);
    wait_dma_and_barrier<NSLOT - 3>();
    read_frags(0, 0, wb[0], gb[0]);
    for (int it = 0; it < nsteps8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cur = u & 1, half = u & 1;
            wait_dma_and_barrier<NSLOT - 6>();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = 3 * u + i + NSLOT;
                issue_slot(it * 8 + q / 3, q % 3, (3 * u + i) % NSLOT);
            }
            read_frags((3 * (u + 1)) % NSLOT, ((u + 1) >> 1) & 1, wb[cur ^ 1], gb[cur ^ 1]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wb[cur][nf][ks], gb[cur][rf][ks], acc[rf][half * 3 + nf]);
#pragma unroll
            for (int i = 0; i < 12; ++i) { if (i < 6) SGB(0x008, 2); else SGB(0x008, 1); SGB(0x100, 1); if (i == 0 || i == 3 || i == 6) SGB(0x010, 1); }
        }
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
    float sum = 0.f;
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) sum += acc[rf][nf][0] + acc[rf][nf][1] + acc[rf][nf][2] + acc[rf][nf][3];
    atomicAdd(out + blockIdx.x, sum);
}

template <int WAVES>
static void run(const __bf16* w, unsigned wbytes, float* out, int nsteps8) {
    typedef void (*kern_t)(const __bf16*, unsigned, float*, int);
    kern_t kern = WAVES == 8 ? ffn_tile_kernel8 : WAVES == 4 ? ffn_tile_kernel4 : ffn_tile_kernel8_k64;
    const int threads = WAVES == 4 ? 256 : 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, 256 * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), LDS, 0, w, wbytes, out, nsteps8);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> h(256);
        hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
        const double steps = nsteps8 * 8.0, flop = steps * 2.0 * 96 * 384 * 32 * 256;
        printf("WAVES %d: %.1f us, %.0f ns/step (%.0f cycles @2.4 GHz; MFMA floor 240 ns), %.0f TFLOP/s, checksum %.6g %s\n", WAVES, ms * 1e3,
               ms * 1e6 / steps, ms * 1e6 / steps * 2.4, flop / ms / 1e9, (double)h[0] + h[255], hipGetErrorString(hipGetLastError()));
    }
}

int main(int argc, char** argv) {
    const int nsteps8 = argc > 1 ? atoi(argv[1]) : 60;  // 480 steps
    const size_t n = (size_t)E * F;
    std::vector<unsigned short> hw(n);
    for (size_t i = 0; i < n; ++i) hw[i] = 0x3c00 + (unsigned short)(((i * 2654435761u) >> 24) & 0x7f);
    __bf16* w;
    float* out;
    hipMalloc(&w, n * 2);
    hipMalloc(&out, 256 * 4);
    hipMemcpy(w, hw.data(), n * 2, hipMemcpyHostToDevice);
    run<8>(w, (unsigned)(n * 2), out, nsteps8);
    run<4>(w, (unsigned)(n * 2), out, nsteps8);
    run<64>(w, (unsigned)(n * 2), out, nsteps8);  // 8 waves, k64 half-column steps
    return 0;
}
