// Dense layer for gfx950 in the parity precision (PP_PREC_F16X3: split-fp16 operands, pp_split.h, three fp16 MFMAs per product),
// twelve-wave form:     out[m, n] = act_fn(sum_k a[m, k] w[n, k] + bias[n]) + residual[r(m), n]
// (every nn.Linear of the ViT-B backbone at 384x288 - qkv, proj, fc1 (+ GELU), fc2; mmpretrain VisionTransformer [3P], call site
// mmpose/models/pose_estimators/base.py:206, ctor args configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67
// with arch 'base'). pp_gemm routes here before the wide-tile kernel (pp_panel_split.hip) when the shape allows it.
//
// The structure is pp_ffn_dma.hip's, applied to a plain GEMM:
//   * a workgroup works on one 192 x 192 output tile at a time, 768 threads = 12 waves, three per SIMD, <= 168 registers;
//   * waves 0-7 compute: wave (rg, cg) = rows 48 rg .., columns 96 cg .. = 3 x 6 fragments (72 accumulator registers); per K-step
//     (32 elements = one 128-byte block per row) one barrier, 18 fragment reads rolling under 54 MFMAs (k_loop_roll); they issue no
//     memory instruction in the loop;
//   * waves 8-11, one per SIMD, issue every buffer_load ... lds piece (a stage = 192 activation rows + 192 weight rows = 48 KiB =
//     48 pieces of 8 rows x 128 B, 12 per wave) two stages ahead on a ring of three, and do the counted vmcnt wait in front of each
//     barrier;
//   * one workgroup per CU walks tiles id, id + CUs, ... (option "linear_loop"; 0: a workgroup per tile) with the ring turning across
//     tiles: the next tile's first two stages are requested under this tile's epilogue;
//   * tiles in bands of four row tiles, row tile fastest: the workgroups resident on one XCD share one row tile per band and half of the
//     weight column tiles (see the kernel);
//   * rows past M: the activation descriptor ends at row M (the DMA writes zeros), the output descriptor too (stores are dropped).
// Three epilogues (template MODE): pp_gemm's, and the two of pp_linear_ln_folded (LayerNorm statistics in / out, residual rows in the
// operand format). History: round 4 ran layers without a residual on a PERSISTENT form whose finished tiles left through the DMA waves
// (qkv 519 - 529 against 549 - 557 us on the wide-tile kernel); with rolling fragment reads, row-pair stores and the tile loop the plain
// kernel passed it (15.4 against 15.6 - 16.2 ms per config 4 step) and the form was removed in round 5 (its DMA waves spilled 66 - 75
// SGPRs, and epilogue VALU shares the SIMDs with the MFMAs whichever wave issues it). Step loop alone (scripts/micro/gemm12.hip,
// M = 55 296): 450 / 583 / 622 us for (N, K) = (2304, 768) / (3072, 768) / (768, 3072).
#include "pp_common.h"
#include "pp_gemm.h"
#include "pp_split.h"

namespace pp {
namespace ldm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 192, BN = 192, CW = 8, WAVES = 12, THREADS = WAVES * 64, NSTAGE = 3;
constexpr int STAGE = (BM + BN) * 128, B_OFF = BM * 128, LDS = NSTAGE * STAGE;
static_assert(LDS <= 160 * 1024, "LDS map");

struct Params {
    const char* a;     // [M, K] split rows
    const char* w;     // [N, K] split rows
    const float* bias; // [N] or NULL
    const float* residual;  // fp32 [*, ldres] or NULL; res_split: [M, N] split rows (may be `out`)
    char* out;         // [M, N]: split rows (out_split) or fp32
    int M, N, K, ldres, res_mod, act, out_split, ntn;
    // LayerNorm folded around the layer (pp_linear_ln_folded): statistics of the activation rows in, of the output rows out
    int res_split;           // residual rows are split rows
    const float* ln_stats;   // [M, K / 96, 2]: per row and 96-column part (mean, sum of squared deviations) of the activation rows, or NULL
    const float* ln_colsum;  // [N]: sum_k w[n, k] (of the split-rounded weights)
    float ln_eps;
    float* stats_out;        // [M, N / 96, 2]: the same statistics of the output rows, or NULL
    float w_inv = 1.0f;      // the weights are stored as w * 2^e (weights.py): accumulators * 2^-e in front of bias / activation / residual (exact)
};

#define LDM_WAITVM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}



// K loop of a computing wave with ROLLING fragment reads (pp_ffn_dma.hip, round 5): the 18 fragment registers of a step are re-read for
// the next stage right behind the last MFMA of this step that uses them - sweeps hi x lo, hi x hi, lo x hi, so the first sweep's operands
// (weights hi, rows lo) are free earliest - and the barrier of stage k + 1 sits inside step k, behind the first row fragment's six MFMAs:
// by then this wave's reads of stage k (issued during step k - 1) have long returned, so the lgkmcnt(0) in front of the barrier costs
// nothing, and what is read behind it comes from stage k + 1, landed. The DMA waves' protocol is unchanged (their barrier k = "stage k has
// landed, stage k - 1 is read out"). `ks` >= 1 steps; stage s sits in ring slot s % NSTAGE.
__device__ __forceinline__ void k_loop_roll(f32x4 (&acc)[3][6], const char* smem, int lane_hi, int lane_lo, int rg, int cg, int ks, int& slot) {
    auto opq = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto rd = [&](int lane_off, int uni, int imm) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (lane_off + uni) + imm); };
    auto bar = []() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));  // lgkmcnt(0): this wave's reads of the stage the barrier frees
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4 ah[3], al[3], bh[6], bl[6];
    int st = slot;  // ring slot of this tile's stage 0 (a workgroup that walks several tiles keeps the ring turning across them)
    bar();  // stage 0 has landed
    {
        const int ua = opq(st * STAGE + rg * 48 * 128), ub = opq(st * STAGE + B_OFF + cg * 96 * 128);
#pragma unroll
        for (int j = 0; j < 6; ++j) bh[j] = rd(lane_hi, ub, j * 2048);
#pragma unroll
        for (int i = 0; i < 3; ++i) al[i] = rd(lane_lo, ua, i * 2048);
#pragma unroll
        for (int i = 0; i < 3; ++i) ah[i] = rd(lane_hi, ua, i * 2048);
#pragma unroll
        for (int j = 0; j < 6; ++j) bl[j] = rd(lane_lo, ub, j * 2048);
    }
    auto step = [&](bool more) {
        const int sn = st + 1 == NSTAGE ? 0 : st + 1;
        const int ua = opq(sn * STAGE + rg * 48 * 128), ub = opq(sn * STAGE + B_OFF + cg * 96 * 128);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {  // hi x lo
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[i][j] = mma(bh[j], al[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
                if (i == 0) bar();  // the barrier of stage k + 1
                if (i >= 1) al[i - 1] = rd(lane_lo, ua, (i - 1) * 2048);
                if (i == 2) al[2] = rd(lane_lo, ua, 2 * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)  // hi x hi
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[i][j] = mma(bh[j], ah[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (more && i == 2) bh[j] = rd(lane_hi, ub, j * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)  // lo x hi
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[i][j] = mma(bl[j], ah[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (more && j == 5) ah[i] = rd(lane_hi, ua, i * 2048);
                if (more && i == 2) bl[j] = rd(lane_lo, ub, j * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
        st = sn;
    };
    for (int k = 0; k + 1 < ks; ++k) step(true);
    step(false);
    slot = st;  // (the slot behind the last stage: the next tile's stage 0)
}

#ifndef LDM_STAMP
#define LDM_STAMP 0  // dev: s_memtime stamps of wave 0 of workgroups 0 and 100 around the K loop and the epilogue of their first tiles (scripts/micro/ldm_stamps.py)
#endif
#ifndef LDM_SKIP_STORES
#define LDM_SKIP_STORES 0  // dev, with LDM_STAMP: only fragment (0, 0) of a tile is stored (wrong results: what does an epilogue cost without its store instructions?)
#endif
#if LDM_STAMP
__device__ unsigned long long g_ldm_stamps[2][64];
#endif
// MODE 0: pp_gemm's epilogue; the two of pp_linear_ln_folded as their own instantiations (one epilogue with every option spills 20 registers):
// MODE 1: LayerNorm statistics of the activation rows IN (qkv / fc1: no residual), MODE 2: residual rows in either format + statistics of the
// output rows OUT (proj / fc2)
// ACT / OUTS / RES / STATS: -1 = read from the parameters at run time (the generic instantiation); a value = the epilogue's switches folded at
// compile time. The generic epilogue is 18 fragments x ~8 wave-uniform branches (150 s_cbranch, 360 v_mov at their joins: ~3 k of its ~12 k
// cycles per tile, round 5 stamps); the shapes a launch plan actually uses get their own instantiation (launch_tile below).
template <int MODE, int ACT = -1, int OUTS = -1, int RES = -1, int STATS = -1>
__global__ __launch_bounds__(THREADS) void linear_dma_kernel(const Params p) {
    const int k_act = ACT < 0 ? p.act : ACT;
    const bool k_out_split = OUTS < 0 ? p.out_split != 0 : OUTS != 0;
    const int k_res = RES < 0 ? (p.residual ? ((MODE == 2 && p.res_split) ? 2 : 1) : 0) : RES;  // 0 none, 1 fp32 rows, 2 split rows
    const bool k_stats = MODE == 2 && (STATS < 0 ? p.stats_out != nullptr : STATS != 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tiles: bands of 4 row tiles, row tile fastest inside a band. Consecutive ids go round the eight XCDs, so XCD x gets row tile x % 4 of
    // every band and the column tiles of parity x / 4: the workgroups resident on an XCD share their activation rows and half of the weights
    // in its L2. A workgroup takes tiles blockIdx.x, + gridDim.x, ...: one each when the grid is the tile count, or - option "linear_loop",
    // grid = the CU count - a column of them with the ring turning ACROSS tiles: the DMA waves request the next tile's first two stages while
    // the computing waves are in this tile's epilogue, so a tile no longer starts with a workgroup launch and two memory round trips
    // (qkv / fc1 / proj of ViT-B have 24 K-steps per tile: ~20 % of a tile's time was not its K loop).
    const int ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntm * p.ntn;
    const int ksteps = p.K / 32;
    auto tile_origin = [&](int t_lin, int& m0, int& n0) {
        const int band = t_lin / (4 * p.ntn), r = t_lin % (4 * p.ntn);
        const int rows_in_band = ntm - band * 4 < 4 ? ntm - band * 4 : 4;
        m0 = (band * 4 + r % rows_in_band) * BM;
        n0 = (r / rows_in_band) * BN;
    };

    if (wv >= CW) {
        // ---------------- DMA waves: a piece is 8 rows x 128 B; lane (row l = lane >> 3, physical chunk lane & 7) fetches logical chunk (lane & 7) ^ l
        const int d = wv - CW;
        const int x_l = lane >> 3;
        const unsigned v = (unsigned)x_l * (unsigned)(p.K * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
        const int row8 = 8 * p.K * 4;
        auto rsrc_a = [&](int m0) {
            const int rows_left = p.M - m0 < BM ? p.M - m0 : BM;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a) + (size_t)m0 * p.K * 4, 0, (unsigned)rows_left * (unsigned)(p.K * 4), 0x00020000);
        };
        auto rsrc_w = [&](int n0) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w) + (size_t)n0 * p.K * 4, 0, (unsigned)BN * (unsigned)(p.K * 4), 0x00020000);
        };
        auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rw, int kk, int st) {
            char* dst = smem + st * STAGE;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + q * 1024), 16, v + (unsigned)(q * row8), kk * 128, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + B_OFF + q * 1024), 16, v + (unsigned)(q * row8), kk * 128, 0, 0);
            }
        };
        // (Measured and removed, round 5: pulling the residual tile towards the L2 from here - 18 extra LDS-DMA pieces into a junk KiB over the last
        //  eight K-steps. The residual fetch costs a launch ~35 us (proj 214 us with it, 177 without: 170 MB at the speed of the HBM interface,
        //  all workgroups asking at the same moment), but 256 tiles x 147 KB do not fit the 32 MB of L2 beside the operand streams: no gain.)
        int t = (int)blockIdx.x, m0, n0;
        tile_origin(t, m0, n0);
        __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rw = rsrc_w(n0);
        issue(ra, rw, 0, 0);
        issue(ra, rw, ksteps > 1 ? 1 : 0, 1);
        int st_i = 2;
        for (; t < ntiles; t += (int)gridDim.x) {
            const bool has_next = t + (int)gridDim.x < ntiles;
            int m1 = m0, n1 = n0;
            if (has_next) tile_origin(t + (int)gridDim.x, m1, n1);
            const __amdgpu_buffer_rsrc_t ra_n = rsrc_a(m1), rw_n = rsrc_w(n1);
            for (int k = 0; k < ksteps; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                LDM_WAITVM(12);  // stage k has landed: this wave's twelve pieces of stage k + 1 may be out
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // stage k + 2 (all computing waves are past their reads of stage k - 1): of this tile, or the next tile's stage 0 / 1, or - behind
                // the last tile - a harmless re-read that keeps the counts
                if (k + 2 < ksteps) issue(ra, rw, k + 2, st_i);
                else if (has_next) issue(ra_n, rw_n, k + 2 - ksteps < ksteps ? k + 2 - ksteps : 0, st_i);
                else issue(ra, rw, 0, st_i);
                st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
            }
            ra = ra_n; rw = rw_n; m0 = m1; n0 = n1;
        }
        LDM_WAITVM(0);
        return;
    }

    // ---------------- computing waves
    const int rg = wv >> 1, cg = wv & 1;
    const int lane_hi = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4), lane_lo = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ (lane & 7)) << 4);
    int slot = 0;  // ring slot of the next tile's stage 0
#if LDM_STAMP
    int n_stamp = 0;
    auto stamp = [&]() {
        if (wv == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
            const unsigned long long ts = __builtin_amdgcn_s_memtime();
            if (lane == 0 && n_stamp < 64) g_ldm_stamps[blockIdx.x == 0 ? 0 : 1][n_stamp] = ts;
            ++n_stamp;
        }
    };
#else
    auto stamp = []() {};
#endif
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
    stamp();  // 3 i: tile i begins
    // (the lane's fragment coordinates from an OPAQUE copy of the lane id, per tile: as loop invariants the epilogue's per-fragment offsets are
    //  hoisted out of the tile loop and held - 150 to 276 spilled registers - through every K loop)
    int ln_ = lane;
    asm volatile("" : "+v"(ln_));
    const int f_row = ln_ & 15, f_kg = ln_ >> 4;
    int m0, n0;
    tile_origin(t, m0, n0);
    const int tn = n0 / BN;
    const int rows_left = p.M - m0 < BM ? p.M - m0 : BM;
    // MODE 1: mean / rstd of this lane's three activation rows from the producer's 96-column parts - fetched HERE, in front of the K loop (the
    // loads fly under the first stages; six registers live through the loop). Behind the loop they are two more memory round trips on a tile's
    // critical path: +3.5 us per tile measured. Eight parts = 768 columns per round, a lane's twelve 16-byte loads in flight at once.
    const int row0 = rg * 48 + f_row;  // + 16 i
    float mu[3] = {0.f, 0.f, 0.f}, rs[3] = {1.f, 1.f, 1.f};
    if (MODE == 1) {
        const int parts = p.K / 96;
        const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_stats) + (size_t)m0 * parts * 2, 0,
                                                                             (unsigned)rows_left * (unsigned)(parts * 8), 0x00020000);
        float sm[3] = {0.f, 0.f, 0.f}, m2[3] = {0.f, 0.f, 0.f};
        if (parts <= 8) {  // one round: mean and squared deviations from the same registers
            f32x4 t[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    t[i][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (2 * u < parts)  // (wave-uniform; parts is even: K % 192 == 0)
                        t[i][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rst, (unsigned)(row0 + i * 16) * (unsigned)(parts * 8), 2 * u * 8, 0));
                }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (2 * u < parts) sm[i] += t[i][u][0] + t[i][u][2];
                mu[i] = sm[i] / (float)parts;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (2 * u < parts) {
                        const float d0 = t[i][u][0] - mu[i], d1 = t[i][u][2] - mu[i];
                        m2[i] += (t[i][u][1] + 96.f * d0 * d0) + (t[i][u][3] + 96.f * d1 * d1);
                    }
            }
        } else {  // wider rows: two passes over the parts
            for (int pass = 0; pass < 2; ++pass) {
                for (int q = 0; q < parts; q += 2) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rst, (unsigned)(row0 + i * 16) * (unsigned)(parts * 8), q * 8, 0));
                        if (pass == 0) {
                            sm[i] += t[0] + t[2];
                        } else {
                            const float d0 = t[0] - mu[i], d1 = t[2] - mu[i];
                            m2[i] += (t[1] + 96.f * d0 * d0) + (t[3] + 96.f * d1 * d1);
                        }
                    }
                }
                if (pass == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) mu[i] = sm[i] / (float)parts;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) rs[i] = 1.0f / sqrtf(m2[i] / (float)p.K + p.ln_eps);  // (rows past M read zeros: their stores are dropped)
    }
    f32x4 acc[3][6];  // [row fragment i][column fragment j]: row 48 rg + 16 i + f_row, columns 96 cg + 16 j + 4 f_kg + (0..3)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    k_loop_roll(acc, smem, lane_hi, lane_lo, rg, cg, ksteps, slot);

    stamp();  // 3 i + 1: K loop done
    // ---------------- epilogue: act_fn(sum + bias) + residual, rows out as split fp16 (two 8-byte halves per lane) or fp32 (16 bytes).
    // Output addressing through a buffer descriptor that ends at row M: the row part of the offset in the VGPR (range-checked), the
    // wave-uniform column part in the scalar offset.
    // With ln_stats the rows of `a` are RAW residual rows and the layer applies the LayerNorm in front of it here (the caller folded gamma
    // into w and beta into the bias):  sum_k (a - mean) rstd w = rstd (acc - mean colsum(w)).  With stats_out the wave leaves (mean, sum of
    // squared deviations) of its 96 columns of every output row for the layer that will do the same with THESE rows.
    const size_t ldo = (size_t)p.N * 4;  // bytes per output row in either format
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)m0 * ldo, 0, (unsigned)rows_left * (unsigned)ldo, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual) + (k_res == 2 ? (size_t)m0 * p.N : 0), 0,
                                                                          k_res == 2 ? (unsigned)rows_left * (unsigned)ldo : 0u, 0x00020000);
    // The residual values of all 18 fragments are fetched FIRST (72 registers: the operand fragments are dead): written inside the store loop
    // each load sits behind the previous fragment's store - `residual` may alias `out`, the compiler must keep that order - and a tile pays 18
    // memory round trips one after the other (measured: ~20 us per tile, as much as the K loop of the proj layer). A lane reads exactly the
    // elements it writes, so the order between DIFFERENT fragments is free.
    constexpr int JG = 6;
    u32x4 resv[3][JG];
#pragma unroll
    for (int i = 0; i < 3; ++i)  // (defined on every path: left to the fetch alone they become values carried round the tile loop, and spill)
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) resv[i][jj] = u32x4{0u, 0u, 0u, 0u};
    auto fetch_residual = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            const int n = n0 + cg * 96 + (j0 + jj) * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int m = m0 + row0 + i * 16;
                resv[i][jj] = u32x4{0u, 0u, 0u, 0u};
                if (k_res == 2) {
                    const unsigned vrow = (unsigned)(row0 + i * 16) * (unsigned)ldo;
                    const int so = (n >> 5) * 128 + (n & 16) * 2;
                    // (row-pair form: the even lane of a pair fetches the hi chunk of both, the odd one the lo chunk; sorted out where they are used)
                    resv[i][jj] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr_s, vrow + (unsigned)(f_kg >> 1) * 16u + (unsigned)(f_kg & 1) * 64u, so, 0));
                } else if (m < p.M) {
                    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                    resv[i][jj] = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(p.residual + (size_t)rr * p.ldres + n + f_kg * 4));
                }
            }
        }
    };
    float rsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (MODE != 1 && k_res != 0 && j % JG == 0) fetch_residual(j);
        const int n = n0 + cg * 96 + j * 16;  // (wave-uniform) first column of the fragment; the lane's four: n + 4 f_kg ..
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n + f_kg * 4);
        if (MODE == 1) cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + n + f_kg * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int m = m0 + row0 + i * 16;
            (void)m;
            f32x4 v;
            if (MODE == 1) {
                const float rsw = rs[i] * p.w_inv;  // (colsum is the sum of the STORED - scaled - weights: the difference carries the scale)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rsw * (acc[i][j][e] - mu[i] * cs[e]) + bv[e];
            } else {
                v = acc[i][j] * p.w_inv + bv;
            }
            if (k_act == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erfc_as(v[e]);
            } else if (k_act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_keep_nan(v[e]);
            }
            const unsigned vrow = (unsigned)(row0 + i * 16) * (unsigned)ldo;
            const int so = (n >> 5) * 128 + (n & 16) * 2;  // split rows: the fragment's 16 hi halves inside their 32-element block (lo: + 64)
            if (MODE == 2 && k_res == 2) {
                const u32x4 rq = resv[i][j % JG];
                const auto s0 = __builtin_amdgcn_permlane16_swap(rq[0], rq[2], false, false);  // -> (own hi, own lo) of values 0, 1
                const auto s1 = __builtin_amdgcn_permlane16_swap(rq[1], rq[3], false, false);  //                         values 2, 3
                const u32x2 rh = {s0[0], s1[0]}, rl = {s0[1], s1[1]};
                const f16x4 h = __builtin_bit_cast(f16x4, rh), l = __builtin_bit_cast(f16x4, rl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)h[e] + (float)l[e];
            } else if (MODE != 1 && k_res == 1) {
                v += __builtin_bit_cast(f32x4, resv[i][j % JG]);  // (rows past M: zeros, their stores are dropped)
            }
            if (k_stats) {
                rsum[i] += (v[0] + v[1]) + (v[2] + v[3]);
                acc[i][j] = v;  // (kept for the second pass)
            }
            if (k_out_split) {
                // (hi, lo) of the four values in eight VALU instructions (split_pair), then the row-pair form (pp_ffn_dma.hip): lanes f_kg, f_kg ^ 1
                // exchange halves - the even one stores the 16-byte hi chunk of both, the odd one the lo chunk: ONE 16-byte store per fragment
                // instead of two 8-byte ones (+ the wait states behind it, see below)
                u32x2 hu, lu;
                { unsigned h_, l_; split_pair(v[0], v[1], h_, l_); hu[0] = h_; lu[0] = l_; }
                { unsigned h_, l_; split_pair(v[2], v[3], h_, l_); hu[1] = h_; lu[1] = l_; }
                const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
                const u32x4 q = {s0[0], s1[0], s0[1], s1[1]};
#if LDM_SKIP_STORES
                if (i != 0 || j != 0) { asm volatile("" ::"v"(q)); continue; }
#endif
                __builtin_amdgcn_raw_buffer_store_b128(q, ro, vrow + (unsigned)(f_kg >> 1) * 16u + (unsigned)(f_kg & 1) * 64u, so, 0);
                asm volatile("s_nop 3" ::"v"(q));
            } else {
                // (one 16-byte store + the wait states the compiler does not insert behind a buffer_store_dwordx4 with an SGPR offset:
                //  its data registers are read a cycle late for lanes 12 - 15 of every row, scripts/micro/mubuf_store_hazard.hip)
                const u32x4 q = __builtin_bit_cast(u32x4, v);
                __builtin_amdgcn_raw_buffer_store_b128(q, ro, vrow + (unsigned)f_kg * 16u, n * 4, 0);
                asm volatile("s_nop 3" ::"v"(q));
            }
        }
    }
    if (k_stats) {
        // two passes (mean, then squared deviations from it): no cancellation whatever the rows' offset is
        const int parts = p.N / 96;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.stats_out + (size_t)m0 * parts * 2, 0, (unsigned)rows_left * (unsigned)(parts * 8), 0x00020000);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float sm = rsum[i];
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            const float mean = sm * (1.0f / 96.f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = acc[i][j][e] - mean;
                    q = __builtin_fmaf(dd, dd, q);
                }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            if (f_kg == 0)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, mean), __builtin_bit_cast(unsigned, q)}, rso,
                                                      (unsigned)(row0 + i * 16) * (unsigned)(parts * 8), (tn * 2 + cg) * 8, 0);
        }
    }
    stamp();  // 3 i + 2: epilogue issued
    }  // tiles of this workgroup
}


// grid of the one-tile kernel: the tile count, or - option "linear_loop" - one workgroup per CU, each walking tiles id, id + CUs, ...
static int loop_grid(int ntiles) {
    if (option("linear_loop") == 0) return ntiles;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return ntiles < cus ? ntiles : cus;
}

}  // namespace ldm

bool linear_dma_supported(const GemmParams& p, int prec, int groups) {
    if (option("linear_dma") == 0) return false;
    if (prec != PP_PREC_F16X3 || groups != 1 || p.gather != G_LINEAR || p.planar_P > 0 || p.ksplit > 1 || p.head_w || p.pool_h > 0) return false;
    if (p.out_bf16 != 0 && p.out_bf16 != 2) return false;
    if (p.K % 32 != 0 || p.K < 64 || p.N % ldm::BN != 0 || p.lda != p.K || p.ldw != p.K || p.ldc != p.N) return false;
    if (p.residual && (p.out_bf16 == 2 || ((p.ldres ? p.ldres : p.ldc) % 4) != 0)) return false;
    if ((size_t)ldm::BM * p.K * 4 >= 0x7ffffff0u || (size_t)ldm::BM * p.N * 4 >= 0x7ffffff0u) return false;
    const long long ntiles = (long long)(p.N / ldm::BN) * ((p.M + ldm::BM - 1) / ldm::BM);
    // two rounds of the chip at least; smaller problems stay with the 128 x 128 / wide-tile kernels. No lower bound on K beyond two stages: at
    // the ViT-S shapes (K = 384; reached when the fused layer kernels are off or the token count is not 192) it is level with the
    // overlapped-epilogue kernel it shadows - 71.5 / 111.2 / 160.8 / 255.7 us against 71.1 / 104.7 / 174.7 / 256.7 us for qkv / fc1 at
    // M = 24 576 / 55 296 (round 5 measurement; the bench script went with pp_linear_ovl.hip in round 6); option "linear_dma" = 0 hands those shapes to the wide-tile kernel (pp_panel_split.hip)
    return ntiles >= 512;
}

int linear_dma_gemm(const GemmParams& g, hipStream_t s) {
    ldm::Params p{};
    p.a = reinterpret_cast<const char*>(g.A);
    p.w = reinterpret_cast<const char*>(g.W);
    p.bias = g.bias;
    p.residual = g.residual;
    p.out = reinterpret_cast<char*>(g.C);
    p.M = g.M; p.N = g.N; p.K = g.K;
    p.ldres = g.ldres ? g.ldres : g.ldc;
    p.res_mod = g.res_mod;
    p.act = g.act;
    p.out_split = g.out_bf16 == 2;
    p.ntn = g.N / ldm::BN;
    p.w_inv = g.w_inv;
    const int grid = p.ntn * ((g.M + ldm::BM - 1) / ldm::BM);
    // pp_gemm's shapes of the ViT plans get their epilogue switches at compile time: qkv (split rows out), fc1 (+ GELU), proj / fc2 / patch embed
    // (fp32 rows out + fp32 residual); anything else the generic instantiation
    void (*kern)(const ldm::Params) = ldm::linear_dma_kernel<0>;
    if (!p.residual && p.out_split && p.act == ACT_NONE) kern = ldm::linear_dma_kernel<0, ACT_NONE, 1, 0>;
    else if (!p.residual && p.out_split && p.act == ACT_GELU) kern = ldm::linear_dma_kernel<0, ACT_GELU, 1, 0>;
    else if (p.residual && !p.out_split && p.act == ACT_NONE) kern = ldm::linear_dma_kernel<0, ACT_NONE, 0, 1>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldm::LDS));
    hipLaunchKernelGGL(kern, dim3(ldm::loop_grid(grid)), dim3(ldm::THREADS), ldm::LDS, s, p);
    PP_LAUNCH_CHECK_AS("linear_dma_tile");
    return PP_OK;
}

}  // namespace pp

// One Linear layer of a ViT block with the LayerNorm in front of it folded in and the statistics of the one behind it emitted (the
// twelve-wave one-tile kernel above; include/probpose_mi355x.h). mmpretrain TransformerEncoderLayer [3P]: x = x + attn(ln1(x));
// x = ffn(ln2(x)) + x - here ln1 / ln2 never run as launches and x lives in the operand format.
extern "C" int pp_linear_ln_folded_supported(int M, int N, int K, int with_ln_stats) {
    if (M <= 0 || N <= 0 || K < 64 || K % 32 != 0 || N % pp::ldm::BN != 0) return 0;
    if (with_ln_stats && (K % 192 != 0)) return 0;  // (statistics come in 96-column parts, two per 192-column tile of their producer)
    if ((size_t)pp::ldm::BM * K * 4 >= 0x7ffffff0u || (size_t)pp::ldm::BM * N * 4 >= 0x7ffffff0u) return 0;
    const long long ntiles = (long long)(N / pp::ldm::BN) * ((M + pp::ldm::BM - 1) / pp::ldm::BM);
    return ntiles >= 512 ? 2 : 1;  // 2: enough tiles for two rounds of the chip (what pp_gemm asks of this kernel); 1: runs, not recommended
}

extern "C" int pp_linear_ln_folded(const void* act, const void* weight, const float* bias, const void* residual, int residual_format,
                                   void* out, int out_format, int M, int N, int K, int act_fn, const float* ln_stats,
                                   const float* ln_colsum, float ln_eps, float* stats_out, void* stream) {
    return pp_linear_ln_folded_ws(act, weight, bias, residual, residual_format, out, out_format, M, N, K, act_fn, ln_stats, ln_colsum, ln_eps, stats_out,
                                  1.0f, stream);
}

extern "C" int pp_linear_ln_folded_ws(const void* act, const void* weight, const float* bias, const void* residual, int residual_format,
                                      void* out, int out_format, int M, int N, int K, int act_fn, const float* ln_stats,
                                      const float* ln_colsum, float ln_eps, float* stats_out, float w_inv_scale, void* stream) {
    using namespace pp;
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_linear_ln_folded: the weight scale must be a power of two in [2^-40, 2^40]");
    }
    PP_REQUIRE(act && weight && out, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: act, weight and out must be non-NULL");
    PP_REQUIRE(M > 0 && N > 0 && K > 0, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: M, N and K must be positive");
    PP_REQUIRE(out_format == PP_OUT_F32 || out_format == PP_OUT_SPLIT, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: out_format is PP_OUT_F32 or PP_OUT_SPLIT");
    PP_REQUIRE(!residual || residual_format == PP_OUT_F32 || residual_format == PP_OUT_SPLIT, PP_ERR_INVALID_ARG,
               "pp_linear_ln_folded: residual_format is PP_OUT_F32 or PP_OUT_SPLIT");
    PP_REQUIRE(act_fn == ACT_NONE || act_fn == ACT_GELU || act_fn == ACT_RELU, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: unknown act_fn");
    PP_REQUIRE(!ln_stats || ln_colsum, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: ln_stats needs ln_colsum");
    PP_REQUIRE(!ln_stats || (!residual && !stats_out), PP_ERR_UNSUPPORTED,
               "pp_linear_ln_folded: a layer with ln_stats (qkv, fc1) takes no residual and emits no statistics - the block has none there");
    PP_REQUIRE(act != out, PP_ERR_INVALID_ARG, "pp_linear_ln_folded: act must not alias out (a tile's rows are read by other workgroups)");
    PP_REQUIRE(pp_linear_ln_folded_supported(M, N, K, ln_stats != nullptr) != 0, PP_ERR_UNSUPPORTED,
               "pp_linear_ln_folded: needs K % 32 == 0 (K % 192 == 0 with ln_stats), K >= 64, N % 192 == 0");
    ldm::Params p{};
    p.a = reinterpret_cast<const char*>(act);
    p.w = reinterpret_cast<const char*>(weight);
    p.bias = bias;
    p.residual = reinterpret_cast<const float*>(residual);
    p.res_split = residual && residual_format == PP_OUT_SPLIT;
    p.out = reinterpret_cast<char*>(out);
    p.M = M; p.N = N; p.K = K;
    p.ldres = N;
    p.res_mod = 0;
    p.act = act_fn;
    p.out_split = out_format == PP_OUT_SPLIT;
    p.ntn = N / ldm::BN;
    p.ln_stats = ln_stats; p.ln_colsum = ln_colsum; p.ln_eps = ln_eps; p.stats_out = stats_out;
    p.w_inv = w_inv_scale;
    const int grid = p.ntn * ((M + ldm::BM - 1) / ldm::BM);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // the layers of the folded plan with their epilogue switches at compile time: qkv / fc1 (statistics in, split rows out), proj / fc2 (split
    // residual in place, statistics out); the first proj (fp32 residual), the last fc2 (fp32 rows out) and anything else: generic
    void (*kern)(const ldm::Params) = ln_stats ? ldm::linear_dma_kernel<1> : ldm::linear_dma_kernel<2>;
    if (ln_stats && p.out_split && p.act == ACT_NONE) kern = ldm::linear_dma_kernel<1, ACT_NONE, 1, 0, 0>;
    else if (ln_stats && p.out_split && p.act == ACT_GELU) kern = ldm::linear_dma_kernel<1, ACT_GELU, 1, 0, 0>;
    else if (!ln_stats && p.res_split && p.out_split && p.act == ACT_NONE && stats_out) kern = ldm::linear_dma_kernel<2, ACT_NONE, 1, 2, 1>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldm::LDS));
    hipLaunchKernelGGL(kern, dim3(ldm::loop_grid(grid)), dim3(ldm::THREADS), ldm::LDS, s, p);
    PP_LAUNCH_CHECK_AS("linear_dma_fold");
    return PP_OK;
}

#if LDM_STAMP
extern "C" int pp_dev_ldm_stamps(unsigned long long* out) {  // dev: 2 x 64 stamps of the last launch (host pointer)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::ldm::g_ldm_stamps), sizeof(unsigned long long) * 128, 0, hipMemcpyDeviceToHost);
}
#endif
