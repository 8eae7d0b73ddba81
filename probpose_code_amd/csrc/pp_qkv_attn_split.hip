// qkv Linear + multi-head self-attention of a ViT layer in ONE launch, in the parity precision (PP_PREC_F16X3: split-fp16
// operands, pp_split.h, three fp16 MFMAs per product), for 192-token sequences with 32-dim heads (ViT-S @ 256x192):
//     qkv = h Wqkv^T + b ;  out[:, head] = softmax(q_head k_head^T * scale) v_head
// (mmpretrain MultiheadAttention.forward [3P]: qkv = self.qkv(x).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4), scaled
// dot-product attention, heads concatenated; the projection that follows lives in pp_ffn_split.hip. Call site
// mmpose/models/pose_estimators/base.py:206, ctor args td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67.)
//
// As two launches (pp_linear_ovl + pp_attention) the (M, 1152) qkv tensor - 113 MB at bs 64 in this 4-byte format - is
// written by one kernel and read by the next, and the Linear layer's tiles leave through a ~8 B/clk/CU store path that is as
// long as its MFMA time. Here a workgroup is one (sequence, head):
//
//   phase 1  [q | k | v](192 x 96) = h_seq(192 x 384) W_head(96 x 384)^T: twelve K-steps on a two-stage ring, a stage = one
//            128-byte block of K per row: 192 token rows + 96 weight rows (the head's 32 q, 32 k and 32 v rows of Wqkv) =
//            36 KiB by LDS-DMA; 8 waves (cg, rg): 48 tokens x 48 outputs each = 3 x 3 fragments, 27 MFMAs per step in two
//            halves around ONE barrier (the loop of pp_panel_split.hip: hi x hi while the lo fragments arrive; the freed
//            buffer takes stage k + 2; lo x hi and hi x lo while the next stage's hi fragments replace the dying ones);
//            the v fragments are computed TRANSPOSED (operands swapped in the MFMA), so that a lane holds four consecutive
//            tokens of one head dim: V^T goes to LDS with 8-byte writes, no 2-byte scatter;
//   phase 2  the ring is overwritten with q, k (raw split blocks, chunk-swizzled) and V^T (a hi and a lo plane) - 73 KiB;
//            twelve 16-query tiles over the eight waves (waves w and w + 4 share a SIMD: three tiles per SIMD):
//            S^T = K Q^T puts a query's scores lane-locally, softmax in registers, P split in registers, O^T = V^T P^T,
//            rows out in the split format (the single-pass form of pp_attention.hip's attention_split_kernel).
//
// 75 KiB of LDS and <= 128 registers: TWO workgroups per CU (four waves per SIMD), so one workgroup's softmax / LDS phase
// runs beside the other's MFMA / fill phase. The twelve heads of a sequence are neighbours on one XCD (block remap), the
// 288 KiB of its LayerNorm rows are fetched from HBM once and hit in that L2 eleven times; Wqkv (1.7 MB) stays in every L2.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {
namespace qka {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef QKA_DBG
#define QKA_DBG 0  // dev ablations (timing only, wrong results): 1 no attention phase, 2 no GEMM phase MFMAs, 4 no DMA (zeros: the MFMAs then run on
                   // zeros too), 8 the fill stops after the ring's first stages (real data stays in LDS: scripts/r06/qka_ablate.py)
#endif
constexpr int DBG = QKA_DBG;

constexpr int S = 192, HD = 32, E = 384, NT = S / 16, THREADS = 512;
constexpr int A_LINES = S, W_LINES = 3 * HD;
constexpr int STAGE = (A_LINES + W_LINES) * 128;  // 36 KiB
constexpr int KB = E / 32;                        // 12 K-steps
constexpr int OFF_Q = 0, OFF_K = S * 128, OFF_V = 2 * S * 128;
constexpr int SPV = S + 4;                        // V^T row pitch in halves: 98 dwords - the 16 rows a ds_read2_b64 group touches fall on 16 distinct bank pairs (mod 32)
constexpr int V_PLANE = HD * SPV * 2;             // bytes of one plane
constexpr int LDS = OFF_V + 2 * V_PLANE;          // 74 240 B
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS >= 2 * STAGE && 2 * LDS <= 160 * 1024, "two workgroups per CU");

struct Params {
    const void* h;      // [n_seq * 192, 384] split: LayerNorm-ed layer input
    const void* w;      // [1152, 384] split: packed qkv weight (rows: q | k | v, head-major inside each)
    const float* bias;  // [1152] or NULL
    void* out;          // [n_seq * 192, 384] split: attention output, heads concatenated
    int n_seq, heads;
    unsigned h_bytes, w_bytes;
    float scale_log2e;
    // LayerNorm folded into the projection (pp_qkv_attention_split_folded): h holds the RAW residual rows (operand format), w / bias carry
    // gamma / beta (weights.fold_layernorm), and mean / rstd of every row come with them
    // The rows are CENTERED (x - mean, the producer pp_proj_ffn_split_folded subtracts the mean it has just computed): LayerNorm(x) W^T + b =
    // rstd ((x - mean) W'^T) + b' with no  - mean * colsum(W')  term - that fp32 difference was the folded form's whole excess error.
    const float* ln_stats;   // [n_seq * 192, 2]: (mean, rstd) per row
    float w_inv;             // the weights are stored as w * 2^e (weights.py): accumulators * 2^-e in front of the bias (exact)
    int q_split;             // 1, or 2 (small launches, deep-ring form): two workgroups per (sequence, head), each projects the whole head and attends for
                             // HALF of the query tiles - the projection is repeated on a CU that would idle, the attention phase halves (13.9 -> ~12 us
                             // per launch at B = 1: a launch of 24 workgroups leaves 232 CUs without work)
};

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (DBG & 2) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// (hi, lo) of four values of an accumulator fragment, regrouped by the row swap of split_store4_rowpair: the even-row lane
// (bit 4 of the lane id clear) gets the 16-byte hi chunk of the pair's eight elements, the odd-row lane the lo chunk
// the value of lane ^ 16 / lane ^ 32 by v_permlane16_swap / v_permlane32_swap (swap(a, a): an even-row lane gets (own, partner's), an odd-row
// lane (partner's, own))
__device__ __forceinline__ float xor16(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto s = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned own_is_first = ((threadIdx.x >> 4) & 1u) == 0u;
    return __builtin_bit_cast(float, own_is_first ? (unsigned)s[1] : (unsigned)s[0]);
}
__device__ __forceinline__ float xor32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned own_is_first = ((threadIdx.x >> 5) & 1u) == 0u;
    return __builtin_bit_cast(float, own_is_first ? (unsigned)s[1] : (unsigned)s[0]);
}
__device__ __forceinline__ u32x4 split_pair16(f32x4 v) {
    u32x2 hu, lu;
    { unsigned h__, l__; split_pair(v[0], v[1], h__, l__); hu[0] = h__; lu[0] = l__; }
    { unsigned h__, l__; split_pair(v[2], v[3], h__, l__); hu[1] = h__; lu[1] = l__; }
    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

#ifndef QKA_STAMP
#define QKA_STAMP 0  // dev: s_memtime stamps of waves 0 and 7 of workgroups 0 and 777 at the phase boundaries (scripts/micro/qka_stamps.py)
#endif
#if QKA_STAMP
__device__ unsigned long long g_qka_stamps[4][16];
#endif
struct TagF { static constexpr bool value = false; };
struct TagT { static constexpr bool value = true; };

// NSTG: stages of the projection's LDS ring. 2 (74 KiB: two workgroups per CU, one's softmax beside the other's fill - the headline batch) or 4
// (144 KiB, one workgroup per CU, three stages requested ahead: the launch of a SMALL batch is 24 - 400 workgroups, each alone on its CU, and
// with one stage ahead its twelve K-steps were twelve memory round trips - 15 us per launch at B = 1 whatever the arithmetic).
template <bool FOLD, int NSTG = 2>
__device__ __forceinline__ void qkv_attention_body(const Params& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv >> 2, rg = wv & 3;  // column half (48 of the 96 outputs), row quarter (48 of the 192 tokens)
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = fr & 7;

#if QKA_STAMP
    int n_stamp = 0;
    auto stamp = [&]() {
        if ((wv == 0 || wv == 7) && (blockIdx.x == 0 || blockIdx.x == 777)) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && n_stamp < 16) g_qka_stamps[(blockIdx.x == 0 ? 0 : 2) + (wv == 0 ? 0 : 1)][n_stamp] = t;
            ++n_stamp;
        }
    };
#else
    auto stamp = []() {};
#endif
    stamp();  // 0: start
    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);  // the heads of a sequence on one XCD, one after the other
    const int qh = p.q_split == 2 ? (id & 1) : 0;  // which half of the query tiles
    if (p.q_split == 2) id >>= 1;
    const int seq = id / p.heads, head = id - seq * p.heads;

    // ---- DMA: 36 instructions of 8 lines per stage (24 token pieces, 12 weight pieces); wave w issues pieces w, w + 8, w + 16
    // (tokens), 24 + w (weight rows of q and k) and, waves 0-3, 32 + w (v rows). Lane (line l = lane >> 3, physical chunk
    // pc = lane & 7) fetches logical chunk pc ^ l: the LDS image is chunk-swizzled by (line & 7).
    const int d_l = lane >> 3;
    const unsigned d_sw = (unsigned)(((lane & 7) ^ d_l) << 4);
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.h), 0, (DBG & 4) ? 0u : p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (DBG & 4) ? 0u : p.w_bytes, 0x00020000);
    const unsigned v_h = (unsigned)(seq * S + 8 * wv + d_l) * (unsigned)(E * 4) + d_sw;                    // + 64 j token rows
    const unsigned v_w0 = (unsigned)((wv >> 2) * E + head * HD + (wv & 3) * 8 + d_l) * (unsigned)(E * 4) + d_sw;  // piece 24 + w
    const unsigned v_w1 = (unsigned)(2 * E + head * HD + (wv & 3) * 8 + d_l) * (unsigned)(E * 4) + d_sw;    // piece 32 + w (w < 4)
    auto issue_stage = [&](int kb, int buf) {
        char* dst = smem + buf * STAGE + wv * 1024;
        const int so = kb * 128;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(h_rsrc, (lds_ptr_t)(dst + j * 8192), 16, v_h + (unsigned)(j * 64 * E * 4), so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + 24 * 1024), 16, v_w0, so, 0, 0);
        if (wv < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + 32 * 1024), 16, v_w1, so, 0, 0);
    };

    // ---- fragment reads: hi halves in 16-byte chunk fg, lo halves in chunk 4 + fg of a line
    const int a_off = (48 * rg + fr) * 128, w_off = A_LINES * 128 + (48 * cg + fr) * 128;
    const int ch_hi = (fg ^ sw) << 4, ch_lo = ((4 + fg) ^ sw) << 4;
    auto frag_a = [&](int buf, int lo, int rf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + a_off + rf * 2048 + (lo ? ch_lo : ch_hi));
    };
    auto frag_w = [&](int buf, int lo, int cf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + w_off + cf * 2048 + (lo ? ch_lo : ch_hi));
    };

    // ================= phase 1: the head's q | k | v rows of the sequence
    // acc[cf][rf]: normal fragments (A = weight rows, B = token rows): lane = token 48 rg + 16 rf + fr, outputs
    // 48 cg + 16 cf + 4 fg + (0..3); transposed ones (column half 1, cf 1 and 2 = the v dims; A = token rows, B = weight rows):
    // lane = v dim 16 (cf - 1) + fr, tokens 48 rg + 16 rf + 4 fg + (0..3)
    f32x4 acc[3][3];
#pragma unroll
    for (int cf = 0; cf < 3; ++cf)
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) acc[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NSTG; ++st) issue_stage(st, st);
    // the first stage has landed: NSTG - 1 younger ones (5 pieces per stage from waves 0-3, 4 from waves 4-7) may fly
    if (wv < 4) wait_vm_lgkm<5 * (NSTG - 1)>(); else wait_vm_lgkm<4 * (NSTG - 1)>();
    __builtin_amdgcn_s_barrier();
    stamp();  // 1: first stage landed
    u32x4 ah[3], wh[3], al[3], wl[3];
#pragma unroll
    for (int cf = 0; cf < 3; ++cf) wh[cf] = frag_w(0, 0, cf);
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) ah[rf] = frag_a(0, 0, rf);

    auto k_loop = [&](auto vt_tag) {
        constexpr bool VT = decltype(vt_tag)::value;  // this wave's fragments cf >= 1 are transposed
        auto mm = [&](int cf, const u32x4& wf, const u32x4& af, f32x4 c) { return (VT && cf >= 1) ? mma(af, wf, c) : mma(wf, af, c); };
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int cb = k % NSTG, nb = (k + 1) % NSTG;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cf = 0; cf < 3; ++cf) wl[cf] = frag_w(cb, 1, cf);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) al[rf] = frag_a(cb, 1, rf);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int cf = 0; cf < 3; ++cf) acc[cf][rf] = mm(cf, wh[cf], ah[rf], acc[cf][rf]);
            __builtin_amdgcn_sched_barrier(0);
            // every wave holds the rest of this stage in registers -> its buffer is free; stage k + 1 has landed once only the stages
            // requested behind it are outstanding (two stages: none)
            {
                const int younger = NSTG == 2 ? 0 : max(0, min(KB - 1, k + NSTG - 1) - (k + 1));  // (a constant after unrolling)
                if (younger == 0) wait_vm_lgkm<0>();
                else if (younger == 1) { if (wv < 4) wait_vm_lgkm<5>(); else wait_vm_lgkm<4>(); }
                else { if (wv < 4) wait_vm_lgkm<10>(); else wait_vm_lgkm<8>(); }
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (k + NSTG < KB && !(DBG & 8)) issue_stage(k + NSTG, cb);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
                for (int cf = 0; cf < 3; ++cf) acc[cf][rf] = mm(cf, wl[cf], ah[rf], acc[cf][rf]);
                if (k + 1 < KB) ah[rf] = frag_a(nb, 0, rf);
            }
#pragma unroll
            for (int cf = 0; cf < 3; ++cf) {
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) acc[cf][rf] = mm(cf, wh[cf], al[rf], acc[cf][rf]);
                if (k + 1 < KB) wh[cf] = frag_w(nb, 0, cf);
            }
        }
    };
    // FOLD: (mean, rstd) of the tokens this lane's fragments belong to. Normal fragments: the lane's token 48 rg + 16 rf + fr; transposed ones
    // (V^T, waves of column half 1): four consecutive tokens 48 rg + 16 rf + 4 fg + (0..3) per fragment row
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    if (cg == 0) k_loop(TagF{}); else k_loop(TagT{});
    // (requested here, all at once: kept through the K loop they cost 6 - 35 spilled registers at the kernel's 128; one pair per fragment inside the
    //  store loop below is a chain of trips to the L2)
    f32x2_t st_n[3] = {f32x2_t{0.f, 1.f}, f32x2_t{0.f, 1.f}, f32x2_t{0.f, 1.f}};
    if (FOLD) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) st_n[rf] = *reinterpret_cast<const f32x2_t*>(p.ln_stats + ((size_t)seq * S + 48 * rg + 16 * rf + fr) * 2);
    }
    f32x4 st_t[3][2];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) st_t[rf][0] = st_t[rf][1] = f32x4{0.f, 1.f, 0.f, 1.f};
    if (FOLD && cg == 1) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            const float* sp = p.ln_stats + ((size_t)seq * S + 48 * rg + 16 * rf + 4 * fg) * 2;
            st_t[rf][0] = *reinterpret_cast<const f32x4*>(sp);
            st_t[rf][1] = *reinterpret_cast<const f32x4*>(sp + 4);
        }
    }
    stamp();  // 2: the twelve K-steps of the qkv projection
    // (every wave passed the last barrier with all its reads of the ring complete: the ring may be overwritten)

    // ---- + bias, out to LDS: q and k as split lines [token][32 hi | 32 lo] (chunk-swizzled like the staged blocks), V^T planes
    {
        char* Qs = smem + OFF_Q;
        char* Ks = smem + OFF_K;
        _Float16* Vh = reinterpret_cast<_Float16*>(smem + OFF_V);
        _Float16* Vl = reinterpret_cast<_Float16*>(smem + OFF_V + V_PLANE);
        const bool odd = (lane & 16) != 0;
#pragma unroll
        for (int cf = 0; cf < 3; ++cf) {
            const int c = 3 * cg + cf;  // 16-column fragment of the 96 outputs: 0, 1 = q; 2, 3 = k; 4, 5 = v
            if (c < 4) {
                const int which = c >> 1, d0 = (c & 1) * 16;  // dims d0 + 4 fg + (0..3)
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + which * E + head * HD + d0 + 4 * fg);
                char* dstb = which == 0 ? Qs : Ks;
                const int chunk = (d0 >> 3) + (fg >> 1);  // the pair (fg, fg ^ 1) fills one 8-dim chunk
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) {
                    const int t = 48 * rg + 16 * rf + fr;
                    // LayerNorm(x) W^T + b = rstd ((x - mean) W'^T) + b' on centered rows (this lane's token); * w_inv: the weights' power-of-two scale
                    const float rs = (FOLD ? st_n[rf][1] : 1.0f) * p.w_inv;
                    const f32x4 val = acc[cf][rf] * rs + bv;
                    const u32x4 q = split_pair16(val);
                    *reinterpret_cast<u32x4*>(dstb + t * 128 + ((((odd ? 4 : 0) + chunk) ^ sw) << 4)) = q;
                }
            } else {
                const int d = (c - 4) * 16 + fr;
                const float bs = p.bias ? p.bias[2 * E + head * HD + d] : 0.f;
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) {
                    const int t0 = 48 * rg + 16 * rf + 4 * fg;
                    f32x4 v = acc[cf][rf] * p.w_inv + bs;
                    if (FOLD) {  // (transposed fragment: the lane's four values are four TOKENS of one v dim)
                        const f32x4 s01 = st_t[rf][0], s23 = st_t[rf][1];
                        v[0] = (s01[1] * p.w_inv) * acc[cf][rf][0] + bs;
                        v[1] = (s01[3] * p.w_inv) * acc[cf][rf][1] + bs;
                        v[2] = (s23[1] * p.w_inv) * acc[cf][rf][2] + bs;
                        v[3] = (s23[3] * p.w_inv) * acc[cf][rf][3] + bs;
                    }
                    u32x2 hv, lv;
                    { unsigned h__, l__; split_pair(v[0], v[1], h__, l__); hv[0] = h__; lv[0] = l__; }
                    { unsigned h__, l__; split_pair(v[2], v[3], h__, l__); hv[1] = h__; lv[1] = l__; }
                    *reinterpret_cast<u32x2*>(Vh + d * SPV + t0) = hv;
                    *reinterpret_cast<u32x2*>(Vl + d * SPV + t0) = lv;
                }
            }
        }
    }
    stamp();  // 3: this wave's q / k / V^T staged
    __syncthreads();
    stamp();  // 4: every wave's
    if (DBG & 1) return;

    // ================= phase 2: attention of the head over the sequence; query tiles w and w + 8
    const char* Qs = smem + OFF_Q;
    const char* Ks = smem + OFF_K;
    const _Float16* Vh = reinterpret_cast<const _Float16*>(smem + OFF_V);
    const _Float16* Vl = reinterpret_cast<const _Float16*>(smem + OFF_V + V_PLANE);
    const int qt_end = p.q_split == 2 ? (qh + 1) * (NT / 2) : NT;
#pragma unroll 1
    for (int qt = (p.q_split == 2 ? qh * (NT / 2) : 0) + wv; qt < qt_end; qt += THREADS / 64) {
        const f16x8 qh = *reinterpret_cast<const f16x8*>(Qs + (qt * 16 + fr) * 128 + ch_hi);
        const f16x8 ql = *reinterpret_cast<const f16x8*>(Qs + (qt * 16 + fr) * 128 + ch_lo);
        // ---- scores: s[kt][i] = q . k for key 16 kt + 4 fg + i of query 16 qt + fr
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            // (at 128 registers per lane the scheduler must not hoist all twelve key tiles' fragments: three tiles in flight)
            if (kt % 3 == 0) __builtin_amdgcn_sched_barrier(0);
            const f16x8 kh = *reinterpret_cast<const f16x8*>(Ks + (kt * 16 + fr) * 128 + ch_hi);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(Ks + (kt * 16 + fr) * 128 + ch_lo);
            s[kt] = split_mma(kh, kl, qh, ql, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- softmax over the 192 keys of this lane's query (fp32)
        float mx = -__builtin_inff();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kt][i]);
        mx = fmaxf(mx, xor16(mx));  // (the gfx950 row swaps - plain VALU - instead of trips through the LDS queue; same pairing, same bits)
        mx = fmaxf(mx, xor32(mx));
        const float mb = mx * p.scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], p.scale_log2e, -mb));  // arg <= 0: raw v_exp_f32
                s[kt][i] = e;
                sum += e;
            }
        sum += xor16(sum);
        sum += xor32(sum);
        // ---- O^T = V^T P^T: the K = 32 block `blk` takes keys 32 blk + 4 fg + (0..3) and 32 blk + 16 + 4 fg + (0..3) per lane -
        // the same permutation of the contraction index on both operands
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int blk = 0; blk < NT / 2; ++blk) {
            if (blk % 2 == 0) __builtin_amdgcn_sched_barrier(0);
            const f32x4 p0 = s[2 * blk], p1 = s[2 * blk + 1];
            u32x4 phu, plu;
            { unsigned h__, l__; split_pair(p0[0], p0[1], h__, l__); phu[0] = h__; plu[0] = l__; }
            { unsigned h__, l__; split_pair(p0[2], p0[3], h__, l__); phu[1] = h__; plu[1] = l__; }
            { unsigned h__, l__; split_pair(p1[0], p1[1], h__, l__); phu[2] = h__; plu[2] = l__; }
            { unsigned h__, l__; split_pair(p1[2], p1[3], h__, l__); phu[3] = h__; plu[3] = l__; }
            const f16x8 ph = __builtin_bit_cast(f16x8, phu), pl = __builtin_bit_cast(f16x8, plu);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int off = (dt * 16 + fr) * SPV + blk * 32 + 4 * fg;
                const u32x2 h0 = *reinterpret_cast<const u32x2*>(Vh + off), h1 = *reinterpret_cast<const u32x2*>(Vh + off + 16);
                const u32x2 l0 = *reinterpret_cast<const u32x2*>(Vl + off), l1 = *reinterpret_cast<const u32x2*>(Vl + off + 16);
                const u32x4 vh = {h0[0], h0[1], h1[0], h1[1]}, vl = {l0[0], l0[1], l1[0], l1[1]};
                o[dt] = split_mma(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, vl), ph, pl, o[dt]);
            }
        }
        // ---- normalise and store: lane holds dims 16 dt + 4 fg + (0..3) of query 16 qt + fr
        const float inv = 1.0f / sum;
        const size_t oidx = ((size_t)seq * S + qt * 16 + fr) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) split_store4_rowpair(p.out, oidx + dt * 16 + 4 * fg, o[dt] * inv, true);
        stamp();  // 5, 6: a query tile done
    }
    stamp();  // last
}

__global__ __launch_bounds__(THREADS, 4) void qkv_attention_split_kernel(const Params p) { qkv_attention_body<false>(p); }
__global__ __launch_bounds__(THREADS, 4) void qkv_attention_split_folded_kernel(const Params p) { qkv_attention_body<true>(p); }
constexpr int LDS_DEEP = 4 * STAGE;  // 144 KiB: the projection's ring of four stages (phase 2 reuses its first 74 KiB)
static_assert(LDS_DEEP >= LDS && LDS_DEEP <= 160 * 1024, "LDS map of the deep-ring form");
__global__ __launch_bounds__(THREADS) void qkv_attention_split_deep_kernel(const Params p) { qkv_attention_body<false, 4>(p); }


// (Round 3's head-PAIR form - one workgroup per (sequence, two heads), a three-stage ring of 48 KiB stages, one workgroup per CU - measured 80.9 -
// 87 us per launch against 77.7 - 79 for this kernel and was retired in round 6; `git log -S qkv_attention_split2_kernel` has it, the numbers are in
// DESIGN_NOTEBOOK.md 4.)
}  // namespace qka

#if QKA_STAMP
extern "C" int pp_dev_qka_stamps(unsigned long long* out) {  // dev: 4 x 16 stamps of the last launch (host pointer)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::qka::g_qka_stamps), sizeof(unsigned long long) * 64, 0, hipMemcpyDeviceToHost);
}
#endif
}  // namespace pp

static int qkv_attention_launch(const void* h_in, const void* wqkv, const float* bqkv, const float* ln_stats, void* out, int n_seq, int seq_len,
                                int heads, int head_dim, float scale, float w_inv_scale, void* stream);

extern "C" int pp_qkv_attention_split_ws(const void* h_in, const void* wqkv, const float* bqkv, void* out, int n_seq, int seq_len, int heads,
                                         int head_dim, float scale, float w_inv_scale, void* stream) {
    return qkv_attention_launch(h_in, wqkv, bqkv, nullptr, out, n_seq, seq_len, heads, head_dim, scale, w_inv_scale, stream);
}

extern "C" int pp_qkv_attention_split(const void* h_in, const void* wqkv, const float* bqkv, void* out, int n_seq, int seq_len,
                                      int heads, int head_dim, float scale, void* stream) {
    return qkv_attention_launch(h_in, wqkv, bqkv, nullptr, out, n_seq, seq_len, heads, head_dim, scale, 1.0f, stream);
}

extern "C" int pp_qkv_attention_split_folded(const void* x_centered, const void* wqkv_folded, const float* bqkv_folded, const float* ln_stats,
                                             void* out, int n_seq, int seq_len, int heads, int head_dim, float scale, float w_inv_scale,
                                             void* stream) {
    PP_REQUIRE(ln_stats && bqkv_folded, PP_ERR_INVALID_ARG, "pp_qkv_attention_split_folded: the row statistics and the folded bias are required");
    return qkv_attention_launch(x_centered, wqkv_folded, bqkv_folded, ln_stats, out, n_seq, seq_len, heads, head_dim, scale, w_inv_scale, stream);
}

static int qkv_attention_launch(const void* h_in, const void* wqkv, const float* bqkv, const float* ln_stats, void* out, int n_seq, int seq_len,
                                int heads, int head_dim, float scale, float w_inv_scale, void* stream) {
    using namespace pp;
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_qkv_attention_split: the weight scale must be a power of two in [2^-40, 2^40]");
    }
    PP_REQUIRE(h_in && wqkv && out, PP_ERR_INVALID_ARG, "pp_qkv_attention_split: NULL argument");
    PP_REQUIRE(n_seq > 0, PP_ERR_INVALID_ARG, "pp_qkv_attention_split: n_seq must be positive");
    PP_REQUIRE(seq_len == qka::S && head_dim == qka::HD && heads * head_dim == qka::E, PP_ERR_UNSUPPORTED,
               "pp_qkv_attention_split: built for 192-token sequences, 12 heads of 32 dims (ViT-S at 256x192)");
    PP_REQUIRE((size_t)n_seq * qka::S * qka::E * 4 < qka::OOB, PP_ERR_UNSUPPORTED, "pp_qkv_attention_split: operand exceeds 2 GiB");
    PP_REQUIRE(h_in != out, PP_ERR_INVALID_ARG, "pp_qkv_attention_split: out must not alias h_in (other heads still read the rows)");
    qka::Params p{};
    p.h = h_in;
    p.w = wqkv;
    p.bias = bqkv;
    p.out = out;
    p.n_seq = n_seq;
    p.heads = heads;
    p.h_bytes = (unsigned)((size_t)n_seq * qka::S * qka::E * 4);
    p.w_bytes = (unsigned)((size_t)3 * qka::E * qka::E * 4);
    p.scale_log2e = scale * 1.44269504088896340736f;
    p.ln_stats = ln_stats;
    p.w_inv = w_inv_scale;
    p.q_split = 1;
    if (ln_stats) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qka::qkv_attention_split_folded_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, qka::LDS));
        hipLaunchKernelGGL(qka::qkv_attention_split_folded_kernel, dim3(n_seq * heads), dim3(qka::THREADS), qka::LDS,
                           reinterpret_cast<hipStream_t>(stream), p);
    } else if (pp::option("qkv_attn_deep") != 0 && n_seq * heads <= 2 * pp_device_cu_count()) {
        // at most two workgroups per CU's worth of launch: nothing to overlap with on a CU, the deep ring hides the memory round trips instead
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qka::qkv_attention_split_deep_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, qka::LDS_DEEP));
        // (measured, ms per replayed step: B = 1 0.730 -> 0.718, B = 2 0.832 -> 0.816, B = 4 0.959 -> 0.951; B = 5 - 240 workgroups - 1.091 -> 1.108)
        if (pp::option("qkv_attn_qsplit") != 0 && 10 * n_seq * heads <= 4 * pp_device_cu_count()) p.q_split = 2;
        hipLaunchKernelGGL(qka::qkv_attention_split_deep_kernel, dim3(p.q_split * n_seq * heads), dim3(qka::THREADS), qka::LDS_DEEP,
                           reinterpret_cast<hipStream_t>(stream), p);
        PP_LAUNCH_CHECK_AS("qkv_attn_deep");
        return PP_OK;
    } else {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(qka::qkv_attention_split_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, qka::LDS));
        hipLaunchKernelGGL(qka::qkv_attention_split_kernel, dim3(n_seq * heads), dim3(qka::THREADS), qka::LDS,
                           reinterpret_cast<hipStream_t>(stream), p);
    }
    PP_LAUNCH_CHECK();
    return PP_OK;
}
