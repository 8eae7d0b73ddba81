// Multi-head self-attention for 432-token sequences (384x288 crops: BASELINE config 4) in the parity precision (PP_PREC_F16X3,
// split-fp16 operands): softmax(q k^T * scale) v per (sequence, head), from the packed (M, 3E) qkv tensor of the split format.
// (mmpretrain MultiheadAttention.forward [3P]: scaled dot-product attention on qkv.reshape(B, N, 3, H, hd); call site
// mmpose/models/pose_estimators/base.py:206; nearest reference config td-hm_ViTPose-base_8xb64-210e_coco-256x192.py:46-61.)
//
// Round 2's kernel for this shape (pp_attention.hip, attention_split_stream_kernel) stages K and V^T of half the keys through
// registers - V with a 2-byte transposing scatter - into 115 KiB of LDS: one workgroup per CU, 337 us per ViT-B layer at bs 32
// for 54 us of MFMA issue. Here K and V stay ROW-MAJOR by key and come in by LDS-DMA:
//   * a workgroup = (sequence, head, query HALF): seven waves, TWO 16-query tiles each (27 tiles = 14 + 13; round 6 - before: query quarters,
//     one tile per wave, 42 % matrix-pipe busy at ViT-B: every K and V^T fragment a wave read fed ONE tile's MFMAs, 32 KiB of LDS reads per
//     wave and 64 keys against 48 MFMAs - the LDS pipe, not the matrix pipe, set the pace. Now a fragment feeds both tiles: half the LDS
//     reads and half the K / V streamed per MFMA); q fragments from global memory once;
//   * the keys stream through a ring of EIGHT 32-key stages (round 6: four, i.e. one pair requested ahead - a pair of stages is ~0.75 us of MFMAs
//     per wave, a trip to HBM / MALL ~2 us: the kernel waited for memory at every pair, 42 % matrix-pipe busy; three pairs ahead now): per stage K and V as HD / 32 sub-tiles of [32 keys][128 B] (the
//     raw split blocks: 32 hi halves | 32 lo halves, 16-byte chunks XOR-swizzled by key & 7 at the source) = 16 KiB at head
//     dim 64; the stages go in PAIRS (one barrier, one running-maximum update and one rescale of O per 64 keys: 149 -> 145 us),
//     three pairs requested ahead; 128 KiB of LDS at head dim 64: one workgroup per CU;
//   * S^T = K Q^T (a query's scores lane-local), online softmax over the stages (running maximum / sum, O rescaled when the
//     maximum moves), P split in registers, O^T = V^T P^T with the V^T fragments read by ds_read_b64_tr_b16 - the gfx950
//     transposing LDS read (within 16 lanes, lane 4 r + q supplies the address of four consecutive halves M[r][4 q ..], lane i
//     receives M[0..3][i]: four keys of one head dim per lane) - for the hi and the lo plane; no register staging, no scatter.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {
namespace adma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int S = 432, NT = S / 16, TQ = 2, QSPLIT = 2, TPW = (NT + QSPLIT * TQ - 1) / (QSPLIT * TQ), THREADS = 64 * TPW;  // 7 waves x 2 tiles
constexpr int SK = 32;                         // keys per stage
constexpr int NSTG = (S + SK - 1) / SK;        // 14 stages (the last one holds 16 keys)
constexpr int RING = 8;                        // stages in the LDS ring: four pairs - the one worked on and three requested behind it

template <int HD>
struct Cfg {
    static constexpr int NB = HD / 32;                   // 128-byte blocks per key and operand
    static constexpr int SUB = SK * 128;                 // one sub-tile: 32 keys x 128 B
    static constexpr int STAGE = 2 * NB * SUB;           // K sub-tiles, then V sub-tiles
    static constexpr int PIECES = STAGE / 1024;          // DMA instructions per stage (8 keys x 128 B each)
    static constexpr int LDS = RING * STAGE;
    static constexpr int DT = HD / 16;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));  // vmcnt(N) only
}

// MINW: waves per SIMD the register allocator is held to - 4 (<= 128 registers: two workgroups per CU, six values spilled at head dim 64) or 1
// (146 registers, one workgroup of seven waves per CU); option "attn_dma_two_wgs" picks, measured in DESIGN.md 5
template <int HD, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void attention_split_dma_kernel(const char* __restrict__ qkv, char* __restrict__ out, int n_seq,
                                                                         int heads, unsigned qkv_bytes, float scale_log2e) {
    using C = Cfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);  // the query quarters and heads of a sequence on one XCD
    const int qs = id % QSPLIT;
    id /= QSPLIT;
    const int seq = id / heads, head = id - seq * heads;
    const int E = heads * HD;
    const unsigned row_bytes = (unsigned)(3 * E * 4);
    const unsigned base = (unsigned)(seq * S) * row_bytes + (unsigned)(head * HD * 4);  // q block 0 of the sequence's first token
    const int qt = (qs * TPW + wv) * TQ;  // this wave's first query tile (it owns qt and qt + 1)
    const bool live = qt < NT;            // (27 tiles over 2 x 7 x 2 slots: the last wave of the second half has one tile)
    const bool live1 = qt + 1 < NT;

    // ---- DMA: piece j of a stage = sub-tile j / 4 (K blocks 0 .. NB - 1, then V blocks), keys 8 (j % 4) .. + 7; lane (l = lane >> 3,
    // pc = lane & 7) fetches the logical 16-byte chunk pc ^ l of key 8 (j % 4) + l
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(qkv), 0, qkv_bytes, 0x00020000);
    const int d_l = lane >> 3;
    const unsigned d_sw = (unsigned)(((lane & 7) ^ d_l) << 4);
    auto issue_stage = [&](int st) {
        char* dst = smem + (st & (RING - 1)) * C::STAGE;
#pragma unroll
        for (int u = 0; u < (C::PIECES + TPW - 1) / TPW; ++u) {
            const int j = wv + u * TPW;
            if (j < C::PIECES) {
                const int sub = j >> 2, kq = j & 3;
                const int which = sub / C::NB, blk = sub - which * C::NB;  // 0 = K, 1 = V
                const unsigned vo = base + (unsigned)(st * SK + kq * 8 + d_l) * row_bytes + (unsigned)((1 + which) * E * 4 + blk * 128) + d_sw;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // ---- q fragments of this wave's tile: lane (query fr, chunk fg) of block g
    f16x8 qh[TQ][C::NB], ql[TQ][C::NB];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        const int tile = qt + t < NT ? qt + t : 0;  // (a tile that does not exist computes on tile 0's queries and stores nothing)
        const char* qrow = qkv + base + (size_t)(tile * 16 + fr) * row_bytes;
#pragma unroll
        for (int g = 0; g < C::NB; ++g) {
            qh[t][g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + fg * 16);
            ql[t][g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + 64 + fg * 16);
        }
    }
    constexpr int NP = NSTG / 2, AHEAD = RING / 2 - 1;  // stage pairs; pairs requested ahead of the one worked on
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) {
        issue_stage(2 * p);
        issue_stage(2 * p + 1);
    }
    const int npw = wv < C::PIECES ? (C::PIECES - wv + TPW - 1) / TPW : 0;  // DMA pieces this wave issues per stage (wave-uniform)

    f32x4 o[TQ][C::DT];
#pragma unroll
    for (int t = 0; t < TQ; ++t)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[TQ] = {-__builtin_inff(), -__builtin_inff()}, l_run[TQ] = {0.f, 0.f};

    const int sw = fr & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the dynamic region (0)
    // V^T fragments through the transposing read: lane fr = 4 r + q of the 16-lane group fg addresses key 4 fg + r, halves 4 q ..
    const int tr_r = fr >> 2, tr_q = fr & 3;
    const int vkey = 4 * fg + tr_r;  // (+ 16 for the second read: same key & 7)

    // Two stages (64 keys) between barriers: one running-maximum update, one rescale of O and one barrier per 48 MFMAs instead
    // of per 24. The pair p + 1 is requested at the top of pair p, into the two slots pair p - 1 has just left.
    static_assert(NSTG % 2 == 0 && RING == 8, "stage pairs on a ring of eight");
    for (int pr = 0; pr < NP; ++pr) {
        __builtin_amdgcn_sched_barrier(0);
        {   // this wave's pieces of pair pr are in once only the pairs requested behind it are outstanding
            // (first pair: everything - the q fragments' plain loads retire out of order with respect to LDS-DMA pieces, so the count says
            //  nothing while they are in flight; from then on only pieces are)
            const int n = pr == 0 ? 0 : min(AHEAD - 1, NP - 1 - pr) * 2 * npw;  // (wave-uniform)
            switch (n) {
                case 0: wait_vm<0>(); break;
                case 2: wait_vm<2>(); break;
                case 4: wait_vm<4>(); break;
                case 6: wait_vm<6>(); break;
                case 8: wait_vm<8>(); break;
                case 12: wait_vm<12>(); break;
                default: wait_vm<0>(); break;
            }
        }
        __builtin_amdgcn_s_barrier();  // every wave's pieces are in; every wave is done with pair pr - 1
        __builtin_amdgcn_sched_barrier(0);
        if (pr + AHEAD < NP) {         // into the two slots pair pr - 1 has just left
            issue_stage(2 * (pr + AHEAD));
            issue_stage(2 * (pr + AHEAD) + 1);
        }
        if (!live) continue;
        const bool tail = pr == NSTG / 2 - 1 && (S % SK) != 0;  // the last stage holds S % 32 = 16 keys: its second key tile does not exist

        // ---- scores of the pair's four key tiles, both query tiles on every K fragment: s[t][kt][i] = q . k for key 64 pr + 16 kt + 4 fg + i of
        // query fr of tile t
        f32x4 sc[TQ][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const char* Ks = smem + ((2 * pr + (kt >> 1)) & (RING - 1)) * C::STAGE;
            f32x4 acc[TQ] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const int r = (kt & 1) * 16 + fr;
#pragma unroll
            for (int g = 0; g < C::NB; ++g) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(Ks + g * C::SUB + r * 128 + ((fg ^ sw) << 4));
                const f16x8 kl = *reinterpret_cast<const f16x8*>(Ks + g * C::SUB + r * 128 + (((4 + fg) ^ sw) << 4));
#pragma unroll
                for (int t = 0; t < TQ; ++t) acc[t] = split_mma(kh, kl, qh[t][g], ql[t][g], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < TQ; ++t) sc[t][kt] = acc[t];
        }
        // ---- online softmax, per query tile
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            if (tail) sc[t][3] = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            float mx = m_run[t];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sc[t][kt][i]);
            {  // max over the four lane groups of a query: the gfx950 row swaps (plain VALU) instead of two trips through the LDS queue
                const unsigned mu = __builtin_bit_cast(unsigned, mx);
                const auto s16 = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
                const unsigned mv = __builtin_bit_cast(unsigned, mx);
                const auto s32 = __builtin_amdgcn_permlane32_swap(mv, mv, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
            }
            // (the running maximum of a query stops moving after its first few key blocks: when it has not moved for ANY query of the tile -
            //  wave-uniform - alpha is exactly 1 for every lane and the rescale of O, 17 multiplications, is skipped)
            if (__builtin_amdgcn_ballot_w64(mx != m_run[t]) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[t] - mx) * scale_log2e);  // (first pair: exp2(-inf) = 0, nothing to rescale)
                l_run[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) o[t][dt] *= alpha;
                m_run[t] = mx;
            }
            const float mb = mx * scale_log2e;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][kt][i], scale_log2e, -mb));  // masked keys: exp2(-inf) = 0
                    sc[t][kt][i] = e;
                    sum += e;
                }
            l_run[t] += sum;  // this lane's keys only; the four lanes of a query are added up at the end
        }
        // ---- O^T += V^T P^T, one K = 32 block per stage of the pair: keys 4 fg + i and 16 + 4 fg + i per lane on both operands; every V^T
        // fragment read serves both query tiles
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int st = 2 * pr + h;
            f16x8 ph[TQ], pl[TQ];
#pragma unroll
            for (int t = 0; t < TQ; ++t) {   // (hi, lo) of the eight probabilities in 16 VALU instructions (split_pair) instead of 32
                u32x4 phu, plu;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned h_, l_;
                    split_pair(sc[t][2 * h][2 * j], sc[t][2 * h][2 * j + 1], h_, l_);
                    phu[j] = h_; plu[j] = l_;
                    split_pair(sc[t][2 * h + 1][2 * j], sc[t][2 * h + 1][2 * j + 1], h_, l_);
                    phu[2 + j] = h_; plu[2 + j] = l_;
                }
                ph[t] = __builtin_bit_cast(f16x8, phu);
                pl[t] = __builtin_bit_cast(f16x8, plu);
            }
            // two head-dim tiles (16 dims each) per group: eight transposing reads in flight, one wait, twelve MFMAs.
            // As asm: the BUILTIN carries no memory operand, so the compiler waits for every LDS-DMA in flight (vmcnt(0)) in front of
            // it - the stages ahead included. The explicit lgkmcnt(0) covers the group's reads.
#pragma unroll
            for (int d2 = 0; d2 < C::DT / 2; ++d2) {
                u32x2 h0[2], h1[2], l0[2], l1[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int dt = 2 * d2 + e;
                    const int g = dt >> 1, c0 = 2 * (dt & 1) + (tr_q >> 1);  // logical hi chunk of dims 16 dt + 4 q ..
                    const unsigned vb = lds_base + (unsigned)((st & (RING - 1)) * C::STAGE + C::NB * C::SUB + g * C::SUB + vkey * 128 + (tr_q & 1) * 8);
                    const unsigned a_hi = vb + (unsigned)((c0 ^ (vkey & 7)) << 4), a_lo = vb + (unsigned)(((4 + c0) ^ (vkey & 7)) << 4);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h0[e]) : "v"(a_hi));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(h1[e]) : "v"(a_hi));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l0[e]) : "v"(a_lo));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(l1[e]) : "v"(a_lo));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0[0]), "+v"(h1[0]), "+v"(l0[0]), "+v"(l1[0]), "+v"(h0[1]), "+v"(h1[1]), "+v"(l0[1]), "+v"(l1[1]));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u32x4 vh = {h0[e][0], h0[e][1], h1[e][0], h1[e][1]}, vl = {l0[e][0], l0[e][1], l1[e][0], l1[e][1]};
#pragma unroll
                    for (int t = 0; t < TQ; ++t)
                        o[t][2 * d2 + e] = split_mma(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, vl), ph[t], pl[t], o[t][2 * d2 + e]);
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        float sum = l_run[t];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        const bool there = t == 0 || live1;  // (wave-uniform; every lane still takes part in the row swaps of the store)
        const size_t oidx = ((size_t)seq * S + (there ? qt + t : qt) * 16 + fr) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) split_store4_rowpair(out, oidx + dt * 16 + 4 * fg, o[t][dt] * inv, there);
    }
}

template <int HD>
static int launch(const void* qkv, void* out, int n_seq, int heads, float scale, hipStream_t s) {
    using C = Cfg<HD>;
    auto kern = option("attn_dma_two_wgs") != 0 ? attention_split_dma_kernel<HD, 4> : attention_split_dma_kernel<HD, 1>;
    const size_t bytes = (size_t)n_seq * S * 3 * heads * HD * 4;
    PP_REQUIRE(bytes < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_attention: qkv tensor exceeds 2 GiB (32-bit buffer offsets)");
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, dim3(n_seq * heads * QSPLIT), dim3(THREADS), C::LDS, s, reinterpret_cast<const char*>(qkv),
                       reinterpret_cast<char*>(out), n_seq, heads, (unsigned)bytes, scale * 1.44269504088896340736f);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace adma

// 432-token sequences in the split format, head dim 32 or 64 (called from pp_attention)
int attention_split_dma(const void* qkv, void* out, int n_seq, int heads, int head_dim, float scale, hipStream_t s) {
    if (head_dim == 64) return adma::launch<64>(qkv, out, n_seq, heads, scale, s);
    if (head_dim == 32) return adma::launch<32>(qkv, out, n_seq, heads, scale, s);
    return fail(PP_ERR_UNSUPPORTED, "pp_attention (LDS-DMA form): head dim 32 or 64");
}

}  // namespace pp
