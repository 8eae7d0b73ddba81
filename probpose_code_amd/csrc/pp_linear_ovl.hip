// Linear layers of PP_PREC_F16X3 (split-fp16 operands, pp_split.h) with the epilogue of tile t under the K-loop of tile t + 1.
//
// pp_panel_split.hip runs these layers (qkv, fc1 of the ViT: K = 384, twelve K-steps per tile) on one persistent workgroup per
// CU; its epilogue - bias, GELU, fp32 -> (hi, lo), 196 KB of stores per tile - runs between two tiles with the matrix pipe
// idle. Measured at bs 64 (scripts/bench_split_gemm.py, PSPLIT_DBG 128 = no epilogue): qkv 74.8 us of which 22 epilogue,
// fc1 121.7 us of which 55 (26 of them the GELU): 0.92 ms of a 6.1 ms step. The stores are bound by the ~8 B/clk/CU a CU can
// push towards HBM (4.2 TB/s chip-wide), the GELU by VALU issue - neither needs the matrix pipe, the L2 -> LDS fill or the
// LDS stage buffers the K-loop lives on.
//
// Here a 192 x 192 tile keeps TWO accumulator sets (2 x 72 registers): while tile t + 1 accumulates, tile t leaves - one
// half-slice (16 rows x 192 columns: row fragment hs >> 1 of row half hs & 1) per K-step:
//     step k       the waves of that row half apply the activation to their 12 values and write them as fp32 into one of two
//                  12 KiB staging buffers;
//     step k + 1   waves 4-7 read the half-slice back as whole rows, split it and store it.
// The K-step's closing barrier orders the two; two staging buffers make the next half-slice's writes safe without another one.
// Roles by wave, because global stores share vmcnt with the LDS-DMA and a vmcnt(0) wait would sit behind them: waves 0-3
// issue ALL the DMA of a stage (12 instructions each) and wait for it, waves 4-7 do ALL the stores and never wait for them.
// The bias sits in LDS for the whole launch (no vector load in the loop). With two accumulator sets 112 registers are left, so
// a stage's fragments are not all held at once: the stage buffer is released at the END of a step (see kstep).
// LDS: 2 stages x 48 KiB + 2 x 12 KiB staging + 12 KiB bias = 132 KiB.
//
// Measured (bs 64, scripts/bench_split_gemm.py): qkv 77.9 -> 71.8 us, fc1 121.1 -> 116.0 us; the f16x3 step one at a time
// 6.71 -> 6.50 ms, with two steps in flight 6.13 -> 6.15 ms (no gain: the other step's kernels already fill the idle matrix
// pipe). Far from the 75 us the K-loop alone would need: with the stores compiled out fc1 still takes 99.7 us (81.6 without the
// GELU) - the GELU of the staging wave is a block of ~1 200 VALU cycles in its instruction stream (interleaving it with the third
// product by sched_group_barrier spilled 64 registers: 147 us), and a storing wave that stalls on the write queue stalls
// its MFMAs with it. 256 registers per lane are the limit of this form.
#include "pp_common.h"
#include "pp_gemm.h"
#include "pp_split.h"

#include <cstdlib>

namespace pp {
namespace lovl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int THREADS = 512, BM = 192, BN = 192, RF = 6, CF = 3;
constexpr int STAGE = (BM + BN) * 128;         // 48 KiB: one 128-byte block of K per row
constexpr int OFF_CST = 2 * STAGE;
constexpr int SLICE = 16 * BN * 4;             // 12 KiB: one half-slice = 16 rows x 192 fp32
constexpr int ROWB = BN * 4;
constexpr int OFF_BIAS = OFF_CST + 2 * SLICE;
constexpr int MAX_N = 3072;
constexpr int LDS = OFF_BIAS + MAX_N * 4;      // 132 KiB
constexpr int NIW = (BM + BN) / 8 / 4;         // DMA instructions per issuing wave and stage (12)
constexpr int JA = BM / 8 / 4;                 // of which activation rows (6)
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS <= 160 * 1024, "LDS");
#ifndef LOVL_DBG
#define LOVL_DBG 0  // dev ablations (timing only, wrong results): 1 no global stores, 2 no staging and no stores, 4 no activation
#endif

template <int VM>
__device__ __forceinline__ void wait_vm_lgkm0() {
    __builtin_amdgcn_s_waitcnt((VM & 15) | (7 << 4) | (0 << 8) | ((VM >> 4) << 14));
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(THREADS, 2) void linear_ovl_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntn * ntm;
    const int nsteps = p.K / 32;
    if ((int)blockIdx.x >= ntiles) return;

    // XCD-aware tile order as in pp_panel_split.hip: the column tiles of a row panel run on one XCD
    auto decode_tile = [&](int t, int& m0, int& n0) {
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        n0 = (t % ntn) * BN;
        m0 = (t / ntn) * BM;
    };

    // ---- bias into LDS, once
    float* lds_bias = reinterpret_cast<float*>(smem + OFF_BIAS);
    for (int i = tid; i < p.N; i += THREADS) lds_bias[i] = p.bias ? p.bias[i] : 0.f;

    // ---- DMA cursor (tile, step), two stages ahead of the MFMAs; waves 0-3 issue
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, p.w_bytes, 0x00020000);
    const int d_l = lane >> 3;
    const unsigned d_kbytes = (unsigned)(((lane & 7) ^ d_l) << 4);
    unsigned a_voff0 = 0, w_voff = 0;  // activation row of DMA instruction j: 32 j rows further; valid while its row < M
    int a_rows_left = 0;
    int i_tile = blockIdx.x, i_step = 0;
    bool i_live = true;
    auto setup_issue_tile = [&]() {
        int m0 = 0, n0 = 0;
        i_live = i_tile < ntiles;
        if (i_live) decode_tile(i_tile, m0, n0);
        const int m = m0 + 8 * cg + d_l;
        a_voff0 = (unsigned)m * (unsigned)(p.lda * 4) + d_kbytes;
        a_rows_left = i_live ? p.M - m : 0;  // instruction j is in bounds iff 32 j < a_rows_left
        w_voff = (unsigned)(n0 + 8 * cg + d_l) * (unsigned)(p.ldw * 4) + d_kbytes;  // n < N: N % BN == 0
        i_step = 0;
    };
    auto issue_stage = [&](int buf) {
        if (rg != 0) return;
        const unsigned kb = (unsigned)(i_step * 128);
        char* dst = smem + buf * STAGE + cg * 1024;
#pragma unroll
        for (int j = 0; j < JA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(dst + j * 4096), 16,
                                                     32 * j < a_rows_left ? a_voff0 + (unsigned)(32 * j) * (unsigned)(p.lda * 4) + kb : OOB, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NIW - JA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + BM * 128 + j * 4096), 16,
                                                     i_live ? w_voff + (unsigned)(j * 32) * (unsigned)(p.ldw * 4) + kb : OOB, 0, 0, 0);
        if (++i_step == nsteps) {
            i_tile += gridDim.x;
            setup_issue_tile();
        }
    };

    // ---- fragment addresses: hi halves in chunk f_kg, lo halves in chunk 4 + f_kg of the row (swizzled by row & 7)
    const int sw = f_row & 7;
    const int a_frag_off = (rg * (BM / 2) + f_row) * 128;
    const int w_frag_off = BM * 128 + (cg * (BN / 4) + f_row) * 128;
    auto frag_a = [&](int buf, int lo, int rf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + (((lo * 4 + f_kg) ^ sw) << 4) + a_frag_off + rf * 2048);
    };
    auto frag_w = [&](int buf, int lo, int cf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + (((lo * 4 + f_kg) ^ sw) << 4) + w_frag_off + cf * 2048);
    };

    if (rg == 0) setup_issue_tile();
    issue_stage(0);
    issue_stage(1);
    if (rg == 0) wait_vm_lgkm0<NIW>(); else wait_vm_lgkm0<63>();  // the first stage has landed; the bias is in LDS
    __builtin_amdgcn_s_barrier();

    f32x4 acc[CF][RF];   // the tile being accumulated
    f32x4 old[CF][RF];   // the tile before it (+ bias), leaving slice by slice
    bool have_old = false;
    int om0 = 0, on0 = 0;
    char* const cst = smem + OFF_CST;

    // Half-slice hs = (row fragment hs >> 1, row half hs & 1) of `old`: 16 rows x 192 columns. The waves of that row half apply
    // the activation and write fp32 into staging buffer `buf`, position hs & 1 ...
    auto stage_half = [&](int hs, int buf) {
        if (rg != (hs & 1) || (LOVL_DBG & 2)) return;
        const int ml = f_row;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const int nl = cg * (BN / 4) + cf * 16 + f_kg * 4;
            f32x4 v = old[cf][hs >> 1];
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (p.act == ACT_GELU && !(LOVL_DBG & 4)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erfc_as(v[j]);
            }
            *reinterpret_cast<f32x4*>(cst + buf * SLICE + ml * ROWB + (((nl >> 2) ^ (ml & 7)) << 4)) = v;
        }
    };
    // ... and waves 4-7 take it out as whole rows: 16 rows x 24 lanes of 8 elements = 384 lane tasks
    auto store_half = [&](int hs, int buf) {
        if (rg == 0 || (LOVL_DBG & 2)) return;
        const int t = tid - 256;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + t;
            const int r = idx / (BN / 8), cl = idx - r * (BN / 8);
            const int ml = r;
            const int m = om0 + (hs & 1) * (BM / 2) + (hs >> 1) * 16 + r;
            if (idx >= 16 * (BN / 8) || m >= p.M + ((LOVL_DBG & 1) ? -p.M - 1 : 0)) continue;
            const char* src = cst + buf * SLICE + ml * ROWB;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (((2 * cl) ^ (ml & 7)) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (((2 * cl + 1) ^ (ml & 7)) << 4));
            const size_t eoff = (size_t)m * p.ldc + on0 + cl * 8;
            if (p.out_bf16 == 2) {
                f16x8 hv, lv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hv[j] = split_hi(v0[j]);
                    lv[j] = split_lo(v0[j], hv[j]);
                    hv[4 + j] = split_hi(v1[j]);
                    lv[4 + j] = split_lo(v1[j], hv[4 + j]);
                }
                char* o = split_addr(p.C, eoff);
                *reinterpret_cast<f16x8*>(o) = hv;
                *reinterpret_cast<f16x8*>(o + 64) = lv;
            } else {
                float* o = reinterpret_cast<float*>(p.C) + eoff;
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            }
        }
    };

    int cb = 0;  // ring buffer of the stage being consumed (runs on across tiles)
    u32x4 ah[RF], wh[CF], wl[CF];

    // One K-step. Two accumulator sets leave 112 registers for everything else, so a stage's fragments are NOT all held at once
    // (pp_panel_split.hip keeps hi and lo of both operands - 72 registers - to free the stage buffer in mid-step): here the
    // buffer is released at the END of the step, the lo row fragments replace the hi ones as those die (48 registers), and the
    // next stage's DMA still has one whole step to land.
    //     acc += Wh Ah ;  acc += Wl Ah, Ah <- Al ;  acc += Wh Al
    // st_hs / ld_hs >= 0: half-slice st_hs of the previous tile is staged (activation, fp32 -> LDS) / half-slice ld_hs, staged one
    // step earlier, is taken out by waves 4-7; the step's closing barrier orders the two, the staging buffers alternate.
    auto kstep = [&](const int st_hs, const int ld_hs) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) wh[cf] = frag_w(cb, 0, cf);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) ah[rf] = frag_a(cb, 0, rf);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) wl[cf] = frag_w(cb, 1, cf);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) acc[cf][rf] = mma(wh[cf], ah[rf], acc[cf][rf]);
        if (ld_hs >= 0 && have_old) store_half(ld_hs, ld_hs & 1);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) acc[cf][rf] = mma(wl[cf], ah[rf], acc[cf][rf]);
            ah[rf] = frag_a(cb, 1, rf);
        }
        // (tried: the third product written out inside the staging wave's branch with sched_group_barrier(1 MFMA, 14 VALU) so that
        // the GELU interleaves with it - 64 registers spilled instead of 14, fc1 116 -> 147 us)
        if (st_hs >= 0 && have_old) stage_half(st_hs, st_hs & 1);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) acc[cf][rf] = mma(wh[cf], ah[rf], acc[cf][rf]);
        __builtin_amdgcn_sched_barrier(0);
        // the stage is consumed; the next one (the only DMA outstanding in waves 0-3) must have landed; waves 4-7 wait for their
        // LDS traffic only - their stores fly on
        if (rg == 0) wait_vm_lgkm0<0>(); else wait_vm_lgkm0<63>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_stage(cb);
        cb ^= 1;
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int m0, n0;
        decode_tile(tile, m0, n0);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) acc[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the twelve half-slices of the previous tile spread over the first twelve K-steps: staged in step k, taken out in step
        // k + 1 (the last one right after the loop); shorter K-loops (>= 6 steps) move two per step
        if (nsteps >= 2 * RF) {
#pragma unroll
            for (int k = 0; k < 2 * RF; ++k) kstep(k, k - 1);
            for (int k = 2 * RF; k < nsteps; ++k) kstep(-1, k == 2 * RF ? 2 * RF - 1 : -1);
            if (nsteps == 2 * RF && have_old) store_half(2 * RF - 1, 1);
        } else {
            for (int k = 0; k < nsteps; ++k) kstep(-1, -1);
            if (have_old) {  // (short K: the previous tile leaves between the tiles, two half-slices per barrier)
#pragma unroll
                for (int ks = 0; ks < RF; ++ks) {
                    stage_half(2 * ks, 0);
                    stage_half(2 * ks + 1, 1);
                    wait_vm_lgkm0<63>();
                    __builtin_amdgcn_s_barrier();
                    store_half(2 * ks, 0);
                    store_half(2 * ks + 1, 1);
                    wait_vm_lgkm0<63>();
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        // hand the finished tile over (+ bias, from LDS)
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + n0 + cg * (BN / 4) + cf * 16 + f_kg * 4);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) old[cf][rf] = acc[cf][rf] + bv;
        }
        have_old = true;
        om0 = m0;
        on0 = n0;
    }
    // ---- the workgroup's last tile leaves on its own (the staging buffer of slice ks was last read two barriers ago)
    wait_vm_lgkm0<63>();
    __builtin_amdgcn_s_barrier();  // waves 4-7 may still be reading the last half-slice of the tile before out of staging buffer 1
#pragma unroll
    for (int ks = 0; ks < RF; ++ks) {
        stage_half(2 * ks, 0);
        stage_half(2 * ks + 1, 1);
        wait_vm_lgkm0<63>();
        __builtin_amdgcn_s_barrier();
        store_half(2 * ks, 0);
        store_half(2 * ks + 1, 1);
        wait_vm_lgkm0<63>();
        __builtin_amdgcn_s_barrier();
    }
    if (rg == 0) wait_vm_lgkm0<0>();  // the out-of-bounds DMAs past the last tile must not outlive the workgroup
}

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace lovl

static bool linear_ovl_enabled() {
    const bool on = option("linear_ovl") != 0;  // dev switch for A/B timing (pp_set_option)
    return on;
}

// split-fp16 Linear layers without residual whose tile count fills the chip: qkv and fc1 of the ViT in the f16x3 mode
bool linear_ovl_supported(const GemmParams& p, int prec, int groups) {
    if (!linear_ovl_enabled() || prec != PP_PREC_F16X3 || p.gather != G_LINEAR || groups != 1) return false;
    if (p.residual || p.planar_P > 0 || p.ksplit > 1 || p.head_w) return false;
    if (p.out_bf16 != 0 && p.out_bf16 != 2) return false;
    if (p.N % lovl::BN != 0 || p.N > lovl::MAX_N || p.K % 32 != 0 || p.K / 32 < lovl::RF) return false;
    if (p.lda % 32 != 0 || p.ldw % 32 != 0 || p.ldc % 32 != 0) return false;
    // K >= 768 (ViT-B): the wide-tile kernel wins since its stage cursor left scratch memory (round 4; ViT-B 384x288 bs 64, two steps in
    // flight: 1 664 - 1 678 crops/s with this kernel for qkv / fc1, 1 706 - 1 708 without) - 24 K-steps amortise its epilogue, and this
    // kernel still spills 15 registers (21 scratch accesses per tile, scripts/scratch_in_loops.py)
    if (p.K >= 768) return false;
    const long long ntiles = (long long)(p.N / lovl::BN) * ((p.M + lovl::BM - 1) / lovl::BM);
    return ntiles >= 2 * 256;  // at least two tiles per CU: with one there is no next tile to hide the epilogue under
}

int linear_ovl_gemm(const GemmParams& p, hipStream_t s) {
    using namespace lovl;
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < OOB && p.w_bytes < OOB, PP_ERR_UNSUPPORTED,
               "pp linear (overlapped epilogue): operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const int ntiles = (p.N / BN) * ((p.M + BM - 1) / BM);
    const int grid = ntiles < device_cus() ? ntiles : device_cus();
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_ovl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(linear_ovl_kernel, dim3(grid), dim3(THREADS), LDS, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp
