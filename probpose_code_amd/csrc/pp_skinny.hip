// Linear layers of SMALL batches in the parity precision (PP_PREC_F16X3), column-parallel:
//     out[m, n] = act_fn((sum_k a[m, k] w[n, k]) * w_inv + bias[n]) + residual[r(m), n]      [ ; h[m, :] = LayerNorm(out[m, :]) ]
// (the nn.Linear layers of mmpretrain's TransformerEncoderLayer [3P] - attn.proj, ffn.layers.0.0, ffn.layers.1 - and the patch-embed projection;
// call site mmpose/models/pose_estimators/base.py:206, reached from the one-image / few-person callers mmpose/apis/inference.py:161-196,
// demo/image_demo.py:36-61, demo/topdown_demo_with_mmdet.py:35-41).
//
// Why another GEMM. The layer kernels of the headline plan (pp_qkv_attn_split.hip, pp_ffn_dma.hip) give a workgroup 96 COMPLETE token rows and
// stream the layer's whole weight set through it: right at bs 64 (256 workgroups), wrong for one crop - 384 rows are 4 workgroups on 256 CUs, each
// pulling 4.6 MB of weights through one CU: 131 us per layer launch whatever the batch (scripts/r06/small_batch_profile.py: 2.06 ms per step at
// B = 1, 76 % of it in that launch). Here the OUTPUT is cut in both directions - 32 x 32, 64 x 64 or 96 x 96 tiles, picked so that the chip has at
// least a workgroup per CU - and every CU streams 1 / (N / BN) of the weights.
//
//   * 256 threads = 4 waves as 2 (rows) x 2 (columns); a wave owns RT x CT MFMA tiles of 16 x 16 (v_mfma_f32_16x16x32_f16, three per product:
//     lo hi + hi lo + hi hi, pp_split.h); the weight fragment is the MFMA's A operand, so a lane ends with 4 consecutive columns of one row;
//   * K in stages of 64 elements (two 128-byte blocks per row) on a ring of seven / four / three LDS stages (32 / 64 / 96-edge tiles) filled by LDS-DMA (`buffer_load ... lds`, 16 bytes per
//     lane, source chunk XOR-swizzled by the row so that fragment reads are conflict-free); rows past M read as zeros (buffer bounds);
//   * epilogue in registers: * w_inv (the weights' power-of-two scale), + bias, GELU (A & S 7.1.26 as everywhere in this mode), + fp32 residual
//     (optionally a table broadcast over the batch: pos_embed), fp32 or split-fp16 rows out;
//   * OPTIONAL LayerNorm tail without a second launch and without a grid barrier: every workgroup counts itself in on its row block's counter when
//     its tile is stored; the one that arrives LAST (it waits for nobody) re-reads the block's complete fp32 rows from L2, normalises them and
//     writes the operand-format rows for the next layer, then puts the counter back to zero. The result does not depend on who arrives last.
//     A launch costs ~2.7 us of dependency gap in a replayed graph: per layer this saves two of six.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {
namespace sk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int THREADS = 256, KS = 2;  // k-blocks (of 32 elements) per stage
// Stages in the LDS ring per tile edge (32 / 64 / 96): what a workgroup has in flight is what it gets per memory round trip (~2 us from HBM / MALL
// at these sizes - the first version waited for one 16 KiB stage at a time and spent 2.4 us per 64 columns of K). 32 x 32 tiles keep six stages in
// flight (a whole K = 384), 64 x 64 three, 96 x 96 two: 112 / 128 / 144 KiB.
// (by the rows a stage holds, BM + BN: 64 -> 7 stages, 96 -> 5, 128 -> 4, 160 / 192 -> 3)
__host__ __device__ constexpr int stages_of(int rows) { return rows <= 64 ? 7 : (rows <= 96 ? 5 : (rows <= 128 ? 4 : 3)); }
constexpr int ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2;

struct Params {
    const char* a;          // [M, K] split rows
    const char* w;          // [N, K] split rows (stored times 2^e)
    const float* bias;      // [N] or NULL
    const float* residual;  // fp32 [M, N] (may be `out`) or [res_mod, N], or NULL
    char* out;              // [M, N]: fp32 or split rows
    int M, N, K, res_mod, act, out_split;
    unsigned a_bytes, w_bytes;
    float w_inv;
    // LayerNorm tail (out is fp32 then)
    const float* gamma;
    const float* beta;
    char* h_out;            // [M, N] split rows
    float eps;
    int* counters;          // one per row block, zero on entry, zero on exit
    int planar_P, n_valid;  // planar_P > 0 (pp_skinny_conv1x1_planar): out is fp32 (rows / planar_P, n_valid, planar_P) planes, columns >= n_valid are padding
    int xr, xc;             // > 0: XCD-aware tile order (xr * xc == 8, row blocks % xr == 0, column tiles % xc == 0); 0: row-major
    // DECONV form (pp_skinny_deconv): `a` is an NHWC map (n_img, H, W, Cin), row m = pixel, K = 4 Cin runs over the 2 x 2 taps of output phase
    // blockIdx.y = 2 py + px (ConvTranspose2d k4 s2 p1: tap (ty, tx) reads pixel (y + ty - 1 + py, x + tx - 1 + px), zeros outside the map);
    // `w` holds the four phase matrices (Cout, 4 Cin) one after the other, `out` the NHWC map (n_img, 2 H, 2 W, Cout = N) in the split format
    int H, Wd, Cin;
};

#ifndef SK_STAMP
#define SK_STAMP 0  // dev: s_memtime stamps of wave 0 of the workgroup that normalises row block 0 (scripts/r06/skinny_stamps.py)
#endif
#if SK_STAMP
__device__ unsigned long long g_sk_stamps[16];
#define SK_T(i) do { if (LN && wave == 0 && mt == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); sk_t[i] = t__; } } while (0)
#else
#define SK_T(i) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// ---- LayerNorm of the rows m0 + wave * BM / 4 ... of the fp32 output (device-scope loads: the rows were written inside this launch, possibly behind
// another XCD's L2), operand-format rows to h_out. LPR lanes per row, G = 64 / LPR rows per group, GC groups requested at once; a lane owns NV8
// pieces of EIGHT consecutive columns: their hi and lo halves leave as one 16-byte store each (this one CU writes the whole row block: with 8-byte
// stores the tail ran at its store path's ~8 B/clk - 3.2 us per 8 rows per wave, stamps in scripts/r06/skinny_stamps.py).
template <int BM, int LPR, int NV8, int GC>
__device__ __forceinline__ void ln_tail_rows(const Params& p, const __amdgpu_buffer_rsrc_t rout, const float* s_gb, int m0, int wave, int lane) {
    constexpr int G = 64 / LPR, RPW = BM / 4;
    static_assert(RPW % (G * GC) == 0, "the groups must tile a wave's rows");
    const float inv_n = 1.0f / (float)p.N;
    const int lr = lane / LPR, lc = lane - lr * LPR;
    for (int r0 = wave * RPW; r0 < (wave + 1) * RPW; r0 += G * GC) {
        f32x4 v[GC][NV8][2];
#pragma unroll
        for (int g = 0; g < GC; ++g) {
            const int m = m0 + r0 + g * G + lr;
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    v[g][i][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m < p.M)
                        v[g][i][hf] = __builtin_bit_cast(
                            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rout, (unsigned)(((size_t)m * p.N + (i * LPR + lc) * 8 + 4 * hf) * 4), 0, 16));
                }
        }
#pragma unroll
        for (int g = 0; g < GC; ++g) {
            const int m = m0 + r0 + g * G + lr;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) sum += (v[g][i][hf][0] + v[g][i][hf][1]) + (v[g][i][hf][2] + v[g][i][hf][3]);
#pragma unroll
            for (int o = LPR >> 1; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * inv_n;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV8; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[g][i][hf][e] - mean;
                        q = __builtin_fmaf(d, d, q);
                    }
#pragma unroll
            for (int o = LPR >> 1; o >= 1; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = 1.0f / sqrtf(q * inv_n + p.eps);
            if (m < p.M) {
#pragma unroll
                for (int i = 0; i < NV8; ++i) {
                    const int col = (i * LPR + lc) * 8;
                    u32x4 hq, lq;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(s_gb + col + 4 * hf), bt = *reinterpret_cast<const f32x4*>(s_gb + 1024 + col + 4 * hf);
                        f32x4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (v[g][i][hf][e] - mean) * rstd * gm[e] + bt[e];
                        { unsigned h__, l__; split_pair(hv[0], hv[1], h__, l__); hq[2 * hf] = h__; lq[2 * hf] = l__; }
                        { unsigned h__, l__; split_pair(hv[2], hv[3], h__, l__); hq[2 * hf + 1] = h__; lq[2 * hf + 1] = l__; }
                    }
                    char* dst = split_addr(p.h_out, (size_t)m * p.N + col);
                    *reinterpret_cast<u32x4*>(dst) = hq;
                    *reinterpret_cast<u32x4*>(dst + 64) = lq;
                }
            }
        }
    }
}

// any N <= 1024 (a multiple of 64): a group of rows per trip
template <int BM>
__device__ __forceinline__ void ln_tail_rows_any(const Params& p, const __amdgpu_buffer_rsrc_t rout, const float* s_gb, int m0, int wave, int lane) {
    const float inv_n = 1.0f / (float)p.N;
    const int LPR = p.N <= 512 ? 8 : 16, G = 64 / LPR;       // lanes per row, rows per group
    const int nv = p.N / (4 * LPR);                            // 16-byte vectors per lane (<= 16)
    const int lr = lane / LPR, lc = lane - lr * LPR;
    for (int r0 = wave * (BM / 4); r0 < (wave + 1) * (BM / 4); r0 += G) {
        const int m = m0 + r0 + lr;
        const bool live = r0 + lr < (wave + 1) * (BM / 4) && m < p.M;
        f32x4 v[16];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < nv && live)
                v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rout, (unsigned)(((size_t)m * p.N + (i * LPR + lc) * 4) * 4), 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        for (int o = LPR >> 1; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mean;
                    q = __builtin_fmaf(d, d, q);
                }
            }
        for (int o = LPR >> 1; o >= 1; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * inv_n + p.eps);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nv && live) {
                const int col = (i * LPR + lc) * 4;
                const f32x4 g = *reinterpret_cast<const f32x4*>(s_gb + col), b = *reinterpret_cast<const f32x4*>(s_gb + 1024 + col);
                f32x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                split_store4(p.h_out, (size_t)m * p.N + col, hv);
            }
    }
}

template <int RT, int CT, bool LN, bool DECONV = false>
__global__ __launch_bounds__(THREADS) void skinny_linear_kernel(const Params p) {
    static_assert(!(LN && DECONV), "the deconvolution form has no LayerNorm tail");
    constexpr int BM = 32 * RT, BN = 32 * CT, ROWS = BM + BN;
    constexpr int NST = stages_of(ROWS), PRE = NST - 1;  // stages in the ring / requested ahead
    constexpr int STAGE = KS * ROWS * 128;
    constexpr int NI = KS * ROWS / 8, NIW = NI / 4;  // DMA instructions per stage (8 rows of one k-block each), per wave
    static_assert(NI % 4 == 0, "DMA instructions must divide over the four waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int ntn = p.N / BN;
    int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    if (!DECONV && p.xr > 0) {
        // XCD-aware tile order. Workgroup b runs on XCD b % 8, each with its own 4 MiB L2: row-major order hands every XCD tiles of every row block
        // and every column tile, so each L2 pulls ALL of A and ALL of W from the memory side (fc2 at B = 1: 4.7 MB per XCD, more than its L2 - the K
        // loop ran at what eight copies of that traffic allow, not at what the ring had in flight). Here the eight XCDs are an xr x xc grid over the
        // tile matrix: an XCD touches 1 / xr of the activation rows and 1 / xc of the weight rows.
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int xri = xcd / p.xc, xci = xcd - xri * p.xc;
        const int tc = ntn / p.xc, tr = (int)(gridDim.x / ntn) / p.xr;  // tiles of an XCD along N and along M
        const int lm = slot / tc;
        mt = xri * tr + lm;
        nt = xci * tc + (slot - lm * tc);
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int nsteps = p.K / (32 * KS);
#if SK_STAMP
    unsigned long long sk_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SK_T(0);

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, p.a_bytes, 0x00020000);
    const int phase = DECONV ? (int)blockIdx.y : 0;  // output phase 2 py + px: its own weight matrix
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w) + (DECONV ? (size_t)phase * p.w_bytes : 0), 0, p.w_bytes, 0x00020000);
    // a DMA instruction moves 8 rows x 128 B: lane (row l = lane >> 3, physical chunk lane & 7) fetches logical chunk (lane & 7) ^ l
    const unsigned d_row = (unsigned)lane >> 3;
    const unsigned d_src = d_row * (unsigned)(p.K * 4) + ((((unsigned)lane & 7u) ^ d_row) << 4);
    // DECONV: the pixel of this lane's activation row in every DMA instruction it takes part in (the row groups a wave serves are the same in
    // every stage): origin offset of pixel (b, y, x) and its coordinates, or y = -100000 for rows past M (every bounds test then fails)
    unsigned pix_off[NIW];
    int pix_y[NIW], pix_x[NIW];
    if constexpr (DECONV) {
#pragma unroll
        for (int u = 0; u < NIW; ++u) {
            const int j = wave + 4 * u;
            const int r = (j % (ROWS / 8)) * 8;
            const int m = m0 + r - BN + (int)d_row;
            const int hw = p.H * p.Wd;
            const int b = m / hw, rem = m - b * hw;
            pix_y[u] = (r >= BN && m < p.M) ? rem / p.Wd : -100000;
            pix_x[u] = rem - (rem / p.Wd) * p.Wd;
            pix_off[u] = (unsigned)m * (unsigned)(p.Cin * 4) + ((((unsigned)lane & 7u) ^ d_row) << 4);
        }
    }
    // DECONV: (tap, channel block) of the k-blocks of the NEXT stage to be requested, advanced by KS per stage - stages are requested in order -
    // instead of an integer division per DMA instruction (first version: ~240 VALU instructions of address arithmetic per wave and stage
    // against 24 MFMAs; 447 us against 143 for the generic kernel at 24 576 rows)
    int nx_tap[KS], nx_cb[KS];
    const int cpb = DECONV ? (p.Cin >> 5) : 1;  // k-blocks per tap
#pragma unroll
    for (int kb = 0; kb < KS; ++kb) {
        nx_tap[kb] = kb / cpb;
        nx_cb[kb] = kb - nx_tap[kb] * cpb;
    }
    auto issue = [&](int s) {
        if (s >= nsteps) return;
        char* dst = smem + (s % NST) * STAGE;
#pragma unroll
        for (int u = 0; u < NIW; ++u) {
            const int j = wave + 4 * u;               // (wave-uniform)
            const int kb = j / (ROWS / 8), r8 = j - kb * (ROWS / 8);
            const int r = r8 * 8;                     // first of the eight rows: < BN weight rows, then activation rows
            const unsigned koff = (unsigned)((s * KS + kb) * 128);
            // (the row part of the address goes into the VGPR offset: the descriptor's range check covers that one, not the scalar offset - rows
            //  past M must fall outside the tensor's extent to read as zeros)
            if (r < BN)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + (kb * ROWS + r) * 128), 16, d_src + (unsigned)((n0 + r) * p.K * 4), koff, 0, 0);
            else if constexpr (DECONV) {
                const int tap = nx_tap[kb], cb = nx_cb[kb];
                const int dy = (tap >> 1) - 1 + (phase >> 1), dx = (tap & 1) - 1 + (phase & 1);
                const int yy = pix_y[u] + dy, xx = pix_x[u] + dx;
                const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
                const unsigned vo = ok ? (unsigned)((int)pix_off[u] + (dy * p.Wd + dx) * p.Cin * 4) : 0x7ffffff0u;  // outside the map: zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + (kb * ROWS + r) * 128), 16, vo, (unsigned)(cb * 128), 0, 0);
            } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + (kb * ROWS + r) * 128), 16, d_src + (unsigned)((m0 + r - BN) * p.K * 4), koff, 0,
                                                         0);
        }
        if constexpr (DECONV) {
#pragma unroll
            for (int kb = 0; kb < KS; ++kb) {
                nx_cb[kb] += KS;
                if (nx_cb[kb] >= cpb) {  // (KS <= Cin / 32)
                    nx_cb[kb] -= cpb;
                    ++nx_tap[kb];
                }
            }
        }
    };

    f32x4 acc[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < PRE; ++s) issue(s);
    // ---- what the epilogue needs from memory, requested NOW (behind the first stages): bias, residual fragments, LayerNorm parameters. Inside the
    // epilogue every one of these loads sat between two stores to `out` - which `residual` may alias, so the compiler must keep the order - and
    // each fragment became its own round trip to the L2 (stamps, scripts/r06/skinny_stamps.py: 5 - 7.6 us of epilogue for six / nine fragments
    // at B = 8, and 5.7 us of LayerNorm tail per group of 8 rows: twelve gamma / beta fetches, each behind the store before it).
    f32x4 bvv[CT], resv[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + 16 * (CT * wn + c) + 4 * f_kg;
        bvv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bvv[c] = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            resv[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!DECONV) {
                const int m = m0 + 16 * (RT * wm + r) + f_row;
                if (p.residual && m < p.M) {
                    const int mr = p.res_mod > 0 ? m % p.res_mod : m;
                    resv[c][r] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mr * p.N + n);
                }
            }
        }
    }
    f32x4 ln_g = {0.f, 0.f, 0.f, 0.f}, ln_b = {0.f, 0.f, 0.f, 0.f};  // thread t: columns 4 t .. 4 t + 3 (N <= 1024)
    if constexpr (LN) {
        if (4 * tid < p.N) {
            ln_g = *reinterpret_cast<const f32x4*>(p.gamma + 4 * tid);
            ln_b = *reinterpret_cast<const f32x4*>(p.beta + 4 * tid);
        }
    }
    const int sw = f_row & 7;
    const int off_hi = f_row * 128 + ((f_kg ^ sw) << 4), off_lo = f_row * 128 + (((4 + f_kg) ^ sw) << 4);
    for (int s = 0; s < nsteps; ++s) {
        // stage s has landed once only the stages requested behind it are outstanding: PRE - 1 of them, fewer near the end of K
        __builtin_amdgcn_sched_barrier(0);
        {
            const int behind = min(PRE - 1, nsteps - 1 - s);  // (workgroup-uniform)
            if (behind >= PRE - 1) wait_vm<(PRE - 1) * NIW>();
            else if (PRE > 2 && behind == 4) wait_vm<(PRE > 4 ? 4 : 0) * NIW>();
            else if (PRE > 2 && behind == 3) wait_vm<(PRE > 3 ? 3 : 0) * NIW>();
            else if (PRE > 2 && behind == 2) wait_vm<(PRE > 2 ? 2 : 0) * NIW>();
            else if (behind == 1) wait_vm<(PRE > 1 ? 1 : 0) * NIW>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(s + PRE);  // into the slot every wave finished reading before this barrier
        const char* st = smem + (s % NST) * STAGE;
#pragma unroll
        for (int kb = 0; kb < KS; ++kb) {
            const char* wb = st + (kb * ROWS + wn * 16 * CT) * 128;
            const char* ab = st + (kb * ROWS + BN + wm * 16 * RT) * 128;
            u32x4 wh[CT], wl[CT], ah[RT], al[RT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                wh[c] = *reinterpret_cast<const u32x4*>(wb + c * 2048 + off_hi);
                wl[c] = *reinterpret_cast<const u32x4*>(wb + c * 2048 + off_lo);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                ah[r] = *reinterpret_cast<const u32x4*>(ab + r * 2048 + off_hi);
                al[r] = *reinterpret_cast<const u32x4*>(ab + r * 2048 + off_lo);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    f32x4 v = acc[c][r];
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[c]), __builtin_bit_cast(f16x8, ah[r]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[c]), __builtin_bit_cast(f16x8, al[r]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[c]), __builtin_bit_cast(f16x8, ah[r]), v, 0, 0, 0);
                    acc[c][r] = v;
                }
        }
    }

    SK_T(1);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)((size_t)p.M * p.N * 4), 0x00020000);
    // ---- epilogue: lane (f_row, f_kg) of tile (c, r) holds row m0 + 16 (RT wm + r) + f_row, columns n0 + 16 (CT wn + c) + 4 f_kg + (0..3)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + 16 * (CT * wn + c) + 4 * f_kg;
        const f32x4 bv = bvv[c];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int m = m0 + 16 * (RT * wm + r) + f_row;
            const bool live = m < p.M;
            f32x4 v = acc[c][r] * p.w_inv + bv;
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erfc_as(v[e]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_keep_nan(v[e]);
            }
            v += resv[c][r];  // (zeros without a residual)
            size_t idx = (size_t)m * p.N + n;
            if constexpr (DECONV) {  // pixel (b, y, x) of the input map -> pixel (2 y + py, 2 x + px) of the output map
                const int hw = p.H * p.Wd;
                const int mm = live ? m : 0;
                const int b = mm / hw, rem = mm - b * hw, y = rem / p.Wd, x = rem - y * p.Wd;
                idx = (((size_t)b * 2 * p.H + 2 * y + (phase >> 1)) * 2 * p.Wd + 2 * x + (phase & 1)) * p.N + n;
            }
            if (!LN && !DECONV && p.planar_P > 0) {  // (image, column, pixel) planes: the 16 rows of a fragment are 16 consecutive pixels of a plane
                if (live) {
                    const int img = m / p.planar_P, pix = m - img * p.planar_P;
                    float* o = reinterpret_cast<float*>(p.out) + ((size_t)img * p.n_valid + n) * p.planar_P + pix;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.n_valid) o[(size_t)e * p.planar_P] = v[e];
                }
            } else if (p.out_split) split_store4_rowpair(p.out, idx, v, live);  // (every lane calls it: row swaps inside)
            else if (LN) {
                // DEVICE-scope store (sc1): the rows are read back inside this launch by a workgroup that may sit on another XCD, i.e. behind another
                // L2. A plain store + an agent-scope release fence does it too - by writing the whole L2 back (buffer_wbl2) and, on the reading side,
                // invalidating one (buffer_inv): measured 26 us per launch at B = 1 against 13 for the same tiles without the tail.
                if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (unsigned)(idx * 4), 0, 16);
            } else if (live) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + idx) = v;
        }
    }

    if constexpr (LN) {
        // ---- LayerNorm tail: the workgroup that completes the row block normalises it
        __shared__ int s_last;
        SK_T(2);
        __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14));  // vmcnt(0): this thread's device-scope stores are acknowledged
        SK_T(3);
        __syncthreads();
        // (every wave is behind its last read of the ring: its first 8 KiB now hold gamma | beta for the tail - LDS reads do not queue behind the
        //  tail's global stores the way the parameter fetches did)
        float* s_gb = reinterpret_cast<float*>(smem);
        if (4 * tid < p.N) {
            *reinterpret_cast<f32x4*>(s_gb + 4 * tid) = ln_g;
            *reinterpret_cast<f32x4*>(s_gb + 1024 + 4 * tid) = ln_b;
        }
        if (tid == 0) {
            const int seen = __hip_atomic_fetch_add(p.counters + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = seen == ntn - 1;
            if (seen == ntn - 1) __hip_atomic_store(p.counters + mt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (everybody of this block has counted)
        }
        __syncthreads();
        SK_T(4);
        if (!s_last) return;  // (the reads below are device-scope loads: they bypass this CU's L1 and this XCD's non-coherent L2 lines)
        // N = 384 (ViT-S) / 768 (ViT-B): six 8-column pieces per lane at 8 / 16 lanes per row, and ALL of the wave's rows requested before anything is
        // reduced - one memory round trip for the block (the first version made one trip per group of 8 rows: two / three dependent trips to the
        // memory side for 64 / 96-row tiles, ~15 us of tail per launch at B = 8)
        if (p.N == 384) ln_tail_rows<BM, 8, 6, RT>(p, rout, s_gb, m0, wave, lane);
        else if (p.N == 768) ln_tail_rows<BM, 16, 6, (RT == 1 ? 2 : RT)>(p, rout, s_gb, m0, wave, lane);
        else ln_tail_rows_any<BM>(p, rout, s_gb, m0, wave, lane);
#if SK_STAMP
        SK_T(5);
        __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14));
        SK_T(6);
        if (wave == 0 && mt == 0 && lane == 0)
            for (int i = 0; i < 7; ++i) g_sk_stamps[i] = sk_t[i];
#endif
    }
}

// Tile SHAPE (32 RT rows x 32 CT columns) for a problem. A launch of these sizes is a few ROUNDS of one workgroup per CU (the rings take 112 -
// 144 KiB), each round one trip through "request the stages - MFMAs - epilogue"; with the LayerNorm tail every round also carries the last
// workgroups' trip through THEIR ROW BLOCKS - 8 rows per wave and memory round trip - which is why launches with a tail prefer FLAT tiles (32
// rows x 96 columns: a third of the tail of a 96 x 96 tile, three times as many row blocks to spread it over). Cost model fitted to
// scripts/r06/skinny_tile_sweep.py (us per launch inside a replayed graph, MI355X, B = 1 .. 24 crops + flip, six shapes x four layer shapes):
//     without tail:  BASE + rounds * ROUND * (0.5 + 0.5 K / 384)
//     with tail:     (FIRST + (rounds - 1) * NEXT) * (1 + 0.2 (K / 384 - 1))          (K = 1536 measured at 1.5 - 1.65x of K = 384)
// e.g. proj + ln2 at B = 8: 41.4 / 25.4 / 28.0 / 25.8 / 29.0 / 21.3 us measured at 32x32 / 64x64 / 96x96 / 32x96 / 32x64 / 64x96, modelled
// 39 / 32 / 27.5 / 25.7 / 32.8 / 20. The results do not depend on the shape (every output sums its K products in the same order), only the time
// does. Option "skinny_tile" = 10 RT + CT (11, 22, 33, 13, 12, 23) forces a shape for A/B timing.
struct Shape {
    int rt, ct;
    double base, round;   // without the LayerNorm tail
    double first, next;   // with it
};
static const Shape SHAPES[] = {{1, 1, 1.5, 3.3, 11.2, 7.0},  {2, 2, 1.0, 6.2, 17.5, 15.0}, {3, 3, 0.0, 11.3, 27.5, 27.0},
                               {1, 3, 1.2, 6.0, 13.2, 12.5}, {1, 2, 1.5, 4.25, 11.8, 10.5}, {2, 3, 0.9, 8.3, 20.0, 19.5},
                               // 96 x 64: the deconvolutions only (groups == 4; deconv2 of one crop + flip: 1 536 pixels x 256 channels x 4 phases = 256 tiles,
                               // ONE round of the chip where 64 x 64 tiles are 384 = two)
                               {3, 2, 0.9, 8.3, 20.0, 19.5}};
static Shape pick_shape(int M, int N, int K, bool tail, int cus, int groups = 1) {
    const int forced = option("skinny_tile");
    for (const Shape& sh : SHAPES)
        if (forced == 10 * sh.rt + sh.ct && N % (32 * sh.ct) == 0 && !(forced == 32 && groups == 1)) return sh;
    const double kf = 0.5 + 0.5 * (double)K / 384.0, kfl = 1.0 + 0.2 * ((double)K / 384.0 - 1.0);
    Shape best = SHAPES[0];
    double best_cost = 1e30;
    for (const Shape& sh : SHAPES) {
        const int bm = 32 * sh.rt, bn = 32 * sh.ct;
        if (N % bn != 0 || (sh.rt == 3 && sh.ct == 2 && groups == 1)) continue;
        const long long wgs = (long long)((M + bm - 1) / bm) * (N / bn) * groups;
        const double rounds = (double)((wgs + cus - 1) / cus);
        const double cost = tail ? (sh.first + (rounds - 1.0) * sh.next) * kfl : sh.base + rounds * sh.round * kf;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = sh;
        }
    }
    return best;
}

}  // namespace sk
}  // namespace pp

#if SK_STAMP
extern "C" int pp_dev_sk_stamps(unsigned long long* out) {  // dev: the stamps of the last launch with a LayerNorm tail (host pointer)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::sk::g_sk_stamps), sizeof(unsigned long long) * 16, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int pp_skinny_linear_tile(int M, int N, int K, int with_layernorm) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 32 != 0) return 0;
    static int cus = 0;
    if (!cus) cus = pp_device_cu_count() > 0 ? pp_device_cu_count() : 256;
    const pp::sk::Shape sh = pp::sk::pick_shape(M, N, K, with_layernorm != 0, cus);
    return 32 * sh.rt * 1000 + 32 * sh.ct;  // rows * 1000 + columns of the tile
}

extern "C" int pp_skinny_linear(const void* act, const void* weight, const float* bias, const float* residual, int res_mod, void* out,
                                int out_format, int M, int N, int K, int act_fn, float w_inv_scale, const float* ln_gamma, const float* ln_beta,
                                float ln_eps, void* ln_out, int* ln_counters, void* stream) {
    using namespace pp;
    PP_REQUIRE(act && weight && out, PP_ERR_INVALID_ARG, "pp_skinny_linear: act, weight and out must be non-NULL");
    PP_REQUIRE(M > 0 && N > 0 && K > 0, PP_ERR_INVALID_ARG, "pp_skinny_linear: M, N and K must be positive");
    PP_REQUIRE(N % 32 == 0 && K % 64 == 0, PP_ERR_UNSUPPORTED, "pp_skinny_linear: needs N % 32 == 0 and K % 64 == 0");
    PP_REQUIRE(out_format == PP_OUT_F32 || out_format == PP_OUT_SPLIT, PP_ERR_INVALID_ARG, "pp_skinny_linear: out_format is PP_OUT_F32 or PP_OUT_SPLIT");
    PP_REQUIRE(act_fn == sk::ACT_NONE || act_fn == sk::ACT_GELU || act_fn == sk::ACT_RELU, PP_ERR_INVALID_ARG, "pp_skinny_linear: unknown act_fn");
    PP_REQUIRE(act != out, PP_ERR_INVALID_ARG, "pp_skinny_linear: act must not alias out (other column tiles still read the rows)");
    PP_REQUIRE(res_mod >= 0 && (!residual || out_format == PP_OUT_F32 || residual != out), PP_ERR_INVALID_ARG, "pp_skinny_linear: bad residual arguments");
    PP_REQUIRE((size_t)M * K * 4 < 0x7ffffff0u && (size_t)N * K * 4 < 0x7ffffff0u && (size_t)M * N * 4 < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_skinny_linear: operands must be smaller than 2 GiB");
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_skinny_linear: the weight scale must be a power of two in [2^-40, 2^40]");
    }
    const bool ln = ln_out != nullptr;
    if (ln) {
        PP_REQUIRE(ln_gamma && ln_beta && ln_counters, PP_ERR_INVALID_ARG, "pp_skinny_linear: the LayerNorm tail needs gamma, beta and the counters");
        PP_REQUIRE(out_format == PP_OUT_F32 && N % 64 == 0 && N <= 1024, PP_ERR_UNSUPPORTED,
                   "pp_skinny_linear: the LayerNorm tail normalises fp32 output rows of at most 1024 columns (a multiple of 64)");
        PP_REQUIRE(ln_out != out && ln_out != act, PP_ERR_INVALID_ARG, "pp_skinny_linear: ln_out must not alias out or act");
    }
    sk::Params p{};
    p.a = reinterpret_cast<const char*>(act);
    p.w = reinterpret_cast<const char*>(weight);
    p.bias = bias;
    p.residual = residual;
    p.out = reinterpret_cast<char*>(out);
    p.M = M; p.N = N; p.K = K;
    p.res_mod = res_mod;
    p.act = act_fn;
    p.out_split = out_format == PP_OUT_SPLIT;
    p.a_bytes = (unsigned)((size_t)M * K * 4);
    p.w_bytes = (unsigned)((size_t)N * K * 4);
    p.w_inv = w_inv_scale;
    p.gamma = ln_gamma; p.beta = ln_beta; p.h_out = reinterpret_cast<char*>(ln_out); p.eps = ln_eps; p.counters = ln_counters;
    const int code = pp_skinny_linear_tile(M, N, K, ln ? 1 : 0);
    const int bm = code / 1000, bn = code % 1000, rt = bm / 32, ct = bn / 32;
    const int grid = ((M + bm - 1) / bm) * (N / bn);
    if (pp::option("skinny_xcd_order") != 0 && ln && M >= 2048) {
        // the split of the eight XCDs over (row blocks, column tiles) that leaves an XCD the fewest distinct operand bytes. Measured (B = 1 .. 16
        // crops + flip, scripts/r06/skinny_tile_sweep.py with PP_OPT_SKINNY_XCD_ORDER 0 / 1): the launches with a LayerNorm tail gain from ~3 000
        // rows on (B = 8: proj 16.3 -> 14.8 us, fc2 29.6 -> 28.7; B = 16: 22.2 -> 20.3, 40.6 -> 38.8), smaller ones and the launches without a
        // tail lose 0.2 - 0.5 us (a workgroup's fill is bounded by what one CU keeps in flight, not by its XCD's distinct bytes)
        const int nmt = (M + bm - 1) / bm, ntn = N / bn;
        double best = 0.0;
        for (int xr = 1; xr <= 8; xr *= 2) {
            const int xc = 8 / xr;
            if (nmt % xr != 0 || ntn % xc != 0) continue;
            const double bytes = (double)M * K / xr + (double)N * K / xc;
            if (p.xr == 0 || bytes < best) {
                best = bytes;
                p.xr = xr;
                p.xc = xc;
            }
        }
    }
    const size_t lds = (size_t)sk::stages_of(bm + bn) * sk::KS * (bm + bn) * 128;
    void (*kern)(const sk::Params) = nullptr;
#define PP_SK_PICK(R, C) \
    if (rt == R && ct == C) kern = ln ? sk::skinny_linear_kernel<R, C, true> : sk::skinny_linear_kernel<R, C, false>
    PP_SK_PICK(1, 1);
    PP_SK_PICK(2, 2);
    PP_SK_PICK(3, 3);
    PP_SK_PICK(1, 3);
    PP_SK_PICK(1, 2);
    PP_SK_PICK(2, 3);
#undef PP_SK_PICK
    PP_REQUIRE(kern != nullptr, PP_ERR_UNSUPPORTED, "pp_skinny_linear: no kernel for the tile shape");
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(sk::THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// 1x1 convolution to a FEW output channels (the final layer of the heatmap branch: 17 keypoint logits, mmpose/models/heads/hybrid_heads/
// probmap_head.py:244-249, 471-472) of a small batch, planar fp32 out (n_img, n_valid, P) - the layout the decode kernel reads: 32 x 32 tiles over
// the pixels (192 workgroups for one crop + flip) instead of the 128 x 128 kernel's 48 (14.3 us at B = 1).
//   act (n_img * P, K) split rows (NHWC map), weight (32 * ceil(n_valid / 32), K) split rows - rows >= n_valid ZERO -, bias of that padded length
extern "C" int pp_skinny_conv1x1_planar(const void* act, const void* weight_padded, const float* bias_padded, float* out, int n_img, int P, int K,
                                        int n_valid, float w_inv_scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(act && weight_padded && out, PP_ERR_INVALID_ARG, "pp_skinny_conv1x1_planar: NULL argument");
    PP_REQUIRE(n_img > 0 && P > 0 && K > 0 && K % 64 == 0 && n_valid > 0, PP_ERR_INVALID_ARG, "pp_skinny_conv1x1_planar: bad shape (K % 64 == 0)");
    const int M = n_img * P, N = 32 * ((n_valid + 31) / 32);
    PP_REQUIRE((size_t)M * K * 4 < 0x7ffffff0u && (size_t)M * n_valid * 4 < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_skinny_conv1x1_planar: operands must be smaller than 2 GiB");
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_skinny_conv1x1_planar: the weight scale must be a power of two in [2^-40, 2^40]");
    }
    sk::Params p{};
    p.a = reinterpret_cast<const char*>(act);
    p.w = reinterpret_cast<const char*>(weight_padded);
    p.bias = bias_padded;
    p.out = reinterpret_cast<char*>(out);
    p.M = M; p.N = N; p.K = K;
    p.act = sk::ACT_NONE;
    p.a_bytes = (unsigned)((size_t)M * K * 4);
    p.w_bytes = (unsigned)((size_t)N * K * 4);
    p.w_inv = w_inv_scale;
    p.planar_P = P;
    p.n_valid = n_valid;
    const int grid = ((M + 31) / 32) * (N / 32);
    const size_t lds = (size_t)sk::stages_of(64) * sk::KS * 64 * 128;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sk::skinny_linear_kernel<1, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sk::skinny_linear_kernel<1, 1, false>), dim3(grid), dim3(sk::THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK_AS("skinny_conv1x1");
    return PP_OK;
}

// ConvTranspose2d(Cin -> Cout, k4, s2, p1, bias=False) + BatchNorm (folded) + ReLU of a SMALL batch: the four output phases as four column-parallel
// GEMMs of one launch (grid y), the activation operand gathered from the NHWC map by LDS-DMA with zeros outside it (mmpose/models/heads/
// hybrid_heads/probmap_head.py:435-472). The generic 128 x 128 kernel runs 24 workgroups at B = 1 (3 x 2 tiles x 4 phases), each with 48 K-steps
// of 48 MFMAs per wave: 55 - 60 us with 90 % of the chip idle; here 32 x 64 tiles are 192 workgroups.
extern "C" int pp_skinny_deconv(const void* act_nhwc, const void* weight, const float* bias, void* out_nhwc, int B, int H, int W, int Cin, int Cout,
                                void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && out_nhwc, PP_ERR_INVALID_ARG, "pp_skinny_deconv: NULL argument");
    PP_REQUIRE(B > 0 && H > 0 && W > 0, PP_ERR_INVALID_ARG, "pp_skinny_deconv: bad shape");
    PP_REQUIRE(Cin > 0 && Cin % 32 == 0 && (4 * Cin) % 64 == 0 && Cout > 0 && Cout % 32 == 0, PP_ERR_UNSUPPORTED,
               "pp_skinny_deconv: needs Cin % 32 == 0 and Cout % 32 == 0");
    const long long M = (long long)B * H * W, K = 4ll * Cin;
    PP_REQUIRE(M * Cin * 4 < 0x7ffffff0ll && (long long)Cout * K * 4 < 0x7ffffff0ll && 4 * M * Cout * 4 < 0x7ffffff0ll, PP_ERR_UNSUPPORTED,
               "pp_skinny_deconv: operands must be smaller than 2 GiB");
    sk::Params p{};
    p.a = reinterpret_cast<const char*>(act_nhwc);
    p.w = reinterpret_cast<const char*>(weight);
    p.bias = bias;
    p.out = reinterpret_cast<char*>(out_nhwc);
    p.M = (int)M; p.N = Cout; p.K = (int)K;
    p.act = sk::ACT_RELU;
    p.out_split = 1;
    p.a_bytes = (unsigned)(M * Cin * 4);
    p.w_bytes = (unsigned)((long long)Cout * K * 4);  // one phase matrix
    p.w_inv = 1.0f;
    p.H = H; p.Wd = W; p.Cin = Cin;
    static int cus = 0;
    if (!cus) cus = pp_device_cu_count() > 0 ? pp_device_cu_count() : 256;
    const sk::Shape sh = sk::pick_shape(p.M, Cout, p.K, false, cus, 4);
    const int bm = 32 * sh.rt, bn = 32 * sh.ct;
    const size_t lds = (size_t)sk::stages_of(bm + bn) * sk::KS * (bm + bn) * 128;
    void (*kern)(const sk::Params) = nullptr;
#define PP_SK_PICK(R, C) \
    if (sh.rt == R && sh.ct == C) kern = sk::skinny_linear_kernel<R, C, false, true>
    PP_SK_PICK(1, 1);
    PP_SK_PICK(2, 2);
    PP_SK_PICK(3, 3);
    PP_SK_PICK(1, 3);
    PP_SK_PICK(1, 2);
    PP_SK_PICK(2, 3);
    PP_SK_PICK(3, 2);
#undef PP_SK_PICK
    PP_REQUIRE(kern != nullptr, PP_ERR_UNSUPPORTED, "pp_skinny_deconv: no kernel for the tile shape");
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(((p.M + bm - 1) / bm) * (Cout / bn)), 4), dim3(sk::THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK_AS("skinny_deconv");
    return PP_OK;
}
