// Fused ViT feed-forward block for gfx950 in the parity precision (PP_PREC_F16X3: split-fp16 operands, pp_split.h, three
// fp16 MFMAs per product):
//     x <- x + GELU(h W1^T + b1) W2^T + b2 ;   h_next <- LayerNorm(x)
// (mmpretrain TransformerEncoderLayer [3P]: x = ffn(ln2(x), identity = x), FFN = Linear - GELU(erf) - Linear, then the next
// layer's ln1 / the final ln1; call site mmpose/models/pose_estimators/base.py:206, ctor args
// configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67).
// As two launches (pp_linear_ovl + pp_gemm_ln) the 4x-wide hidden activation - 151 MB at bs 64 in this 4-byte format - is
// written to HBM and read back in every layer. Here it never leaves the CU.
//
// Why 96 rows per workgroup and a STREAMED row operand (and not 48 resident rows). In this format an operand element is
// 4 bytes, so the 96 x 384 input rows are 144 KiB: they do not fit beside a weight ring. 48 resident rows would fit
// (72 KiB), but every workgroup streams all 4.5 MiB of W1 / W2 whatever its row count, so halving the rows doubles the
// L2 -> LDS fill per FLOP: 56 B/clk/CU at full MFMA rate against the ~34 B/clk a CU can fill at. With 96 rows
// the x k-blocks are re-streamed for each of the 12 hidden chunks next to the W1 blocks (+1.7 MB, L2-resident): 38.8 B/clk.
//
//   * one workgroup owns 96 complete token rows (one per CU at bs 64 with flip test), 512 threads = 8 waves,
//     wave (rg, cg): rows 48 rg .. +47, column quarter cg;
//   * the hidden layer runs in chunks of 128 units; per chunk twenty steps on a ring of FOUR 28 KiB slots:
//       A-step kb (12 per chunk)   P += x[:, kb] W1[chunk, kb]^T   slot = W1 block (128 lines x 128 B) + x block (96 lines);
//                                  wave tile 48 rows x 32 units, 18 MFMAs
//       B-step (j, half) (8)       acc[:, half] += G[:, j] W2[half, chunk j]^T   slot = W2 half block (192 lines);
//                                  wave tile 48 rows x 48 outputs, 27 MFMAs; the G fragments stay for both halves
//     software-pipelined across chunks like pp_mlp.hip: the loop body is [A-steps of chunk c + 1 | B-steps of chunk c] and
//     the GELU of chunk c (fp32 -> erfc form -> (hi, lo) -> LDS tile G, 48 KiB) is spread over the A-steps of chunk c + 1;
//   * every step is [hi x hi products while the lo fragments are read | ONE barrier | the two cross products while the next
//     step's hi fragments are read]: at the barrier the step's slot is free again (all of it has been read) and the next
//     step's slot must have landed, so the DMA of step s + 4 goes into the slot of step s: three steps (up to 84 KiB) in flight;
//   * W1 / W2 come PRE-PACKED in consumption order (pp_ffn_split_pack_weights: per chunk 12 W1 blocks of 16 KiB, then
//     8 W2 half blocks of 24 KiB, 128-byte lines with the LDS XOR swizzle already applied), so a weight DMA instruction
//     is a linear 1 KiB copy; the x lines are 128-byte segments of the row-major split tensor, swizzled at the source;
//   * the 96 x 384 accumulators start from residual + b2 (loaded under the first chunk's A-steps) and end in the LayerNorm
//     epilogue (row statistics in registers, one LDS exchange between the column quarters).
// LDS: 48 KiB G + 4 x 28 KiB ring = 160 KiB.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {
namespace ffs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef FFS_DBG
#define FFS_DBG 0  // dev ablations (timing only, wrong results): 2 no GELU, 4 no MFMA, 8 no DMA, 16 no fragment reads
#endif
constexpr int DBG = FFS_DBG;
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_VALU = 0x002, SG_MFMA = 0x008, SG_VMEM = 0x010, SG_DS_READ = 0x100, SG_DS_WRITE = 0x200;

constexpr int BM = 96, E = 384, CHUNK = 128, THREADS = 512;
constexpr int KB = E / 32;                       // 12 k-blocks of the input width
constexpr int G_KB = BM * 128;                   // 12 KiB: 96 rows x one 128-byte block
constexpr int OFF_G = 0;                         // [4][96][128 B]
constexpr int OFF_RING = 4 * G_KB;               // 48 KiB
constexpr int SLOTB = 28 * 1024, NSLOT = 4;
constexpr int LDS = OFF_RING + NSLOT * SLOTB;    // 163 840 B
constexpr int X_OFF = 16 * 1024;                 // A slot: the x lines sit behind the 128 weight lines
constexpr int NA = KB, NB = 8, STEPS = NA + NB;  // steps per chunk
constexpr int A_BLOCK = CHUNK * 128;             // 16 KiB
constexpr int B_BLOCK = (E / 2) * 128;           // 24 KiB
constexpr int B_PART = NA * A_BLOCK;             // 192 KiB: offset of the W2 half blocks inside a chunk's stream
constexpr int CHUNK_BYTES = B_PART + NB * B_BLOCK;  // 384 KiB
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS == 160 * 1024, "LDS map");
static_assert(NA % NSLOT == 0 && STEPS % NSLOT == 0, "ring positions must repeat per chunk");

struct Params {
    const void* h;         // [M, 384] split: LayerNorm-ed block input
    const void* wpack;     // pre-packed W1 / W2 stream (pp_ffn_split_pack_weights)
    const float* b1;       // [F]
    const float* b2;       // [384]
    const float* residual; // fp32 [M, 384] (may alias x_out)
    float* x_out;          // fp32 [M, 384]
    const float* gamma;
    const float* beta;
    void* h_out;           // [M, 384] split: LayerNorm(x_out) (may alias h)
    int M, F;
    unsigned h_bytes, w_bytes;
    float eps;
};

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (DBG & 4) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// DMA instructions of one step by wave half: A-steps 16 W1 pieces by waves 0-3 (4 each) + 12 x pieces by waves 4-7 (3 each),
// B-steps 24 pieces, 3 per wave
__host__ __device__ constexpr bool is_a(int t) { return ((t % STEPS) + STEPS) % STEPS < NA; }
__host__ __device__ constexpr int n_ops(int t, int rg) { return is_a(t) ? (rg == 0 ? 4 : 3) : 3; }

template <int N0, int N1>
__device__ __forceinline__ void wait_and_barrier(int rg) {
    // vmcnt(N) lgkmcnt(0) as a builtin (the compiler's wait-count bookkeeping sees it), N by wave half; then the barrier
    static_assert(N0 >= 0 && N0 < 64 && N1 >= 0 && N1 < 64, "vmcnt immediate");
    __builtin_amdgcn_sched_barrier(0);
    if (rg == 0) __builtin_amdgcn_s_waitcnt((N0 & 15) | (7 << 4) | (0 << 8) | ((N0 >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt((N1 & 15) | (7 << 4) | (0 << 8) | ((N1 >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(THREADS, 2) void ffn_split_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nchunks = p.F / CHUNK;
    // every workgroup walks the hidden chunks in a different rotation (rank inside the XCD): all CUs stream the SAME weights
    const int c_rot = (int)(blockIdx.x >> 3) % nchunks;
    auto chunk_of = [&](int i) { const int c = i + c_rot; return c >= nchunks ? c - nchunks : c; };

    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.h), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpack), 0, p.w_bytes, 0x00020000);
    char* const ring = smem + OFF_RING;

    // ---- DMA addressing. A weight piece is a linear 1 KiB copy (lane i -> byte 16 i). An x piece is 8 rows x 128 B: lane
    // (row l = lane >> 3, physical chunk pc = lane & 7) fetches logical chunk pc ^ l (rows 8 p + l: (row & 7) == l).
    const unsigned w_lane = (unsigned)lane * 16u;
    const int x_l = lane >> 3;
    const unsigned x_lane = (unsigned)(((lane & 7) ^ x_l) << 4);
    // step t of the iteration that handles chunk index `ci` as its A-chunk and `ci - 1` as its B-chunk
    auto issue_a = [&](int ci, int kb, int slot) {
        if (DBG & 8) return;
        const bool live = ci < nchunks;
        char* dst = ring + slot * SLOTB;
        if (rg == 0) {
            const int so = live ? chunk_of(ci) * CHUNK_BYTES + kb * A_BLOCK : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pi = 4 * wv + u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + pi * 1024), 16, live ? w_lane + (unsigned)(pi * 1024) : OOB, so, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int pi = 3 * (wv - 4) + u;
                const int m = m0 + 8 * pi + x_l;
                const unsigned vo = (unsigned)m * (unsigned)(E * 4) + x_lane;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(h_rsrc, (lds_ptr_t)(dst + X_OFF + pi * 1024), 16, (live && m < p.M) ? vo : OOB, kb * 128, 0, 0);
            }
        }
    };
    auto issue_b = [&](int ci, int s, int slot) {
        if (DBG & 8) return;
        const bool live = ci >= 0 && ci < nchunks;
        char* dst = ring + slot * SLOTB;
        const int so = live ? chunk_of(ci) * CHUNK_BYTES + B_PART + s * B_BLOCK : 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int pi = 3 * wv + u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + pi * 1024), 16, live ? w_lane + (unsigned)(pi * 1024) : OOB, so, 0, 0);
        }
    };
    // step index t relative to the start of iteration `it` (which runs A-steps of chunk it + 1 and B-steps of chunk it);
    // t < 0: the peeled A-steps of chunk 0 (t = -12 .. -1), t >= 20: the next iteration
    auto issue_step = [&](int it, int t) {
        const int slot = (t + 4 * STEPS) & (NSLOT - 1);
        if (t < 0) issue_a(0, t + NA, slot);
        else if (t < NA) issue_a(it + 1, t, slot);
        else if (t < STEPS) issue_b(it, t - NA, slot);
        else issue_a(it + 2, t - STEPS, slot);
    };

    // ---- fragment reads: hi halves in 16-byte chunk f_kg, lo halves in chunk 4 + f_kg of a line, swizzled by line & 7
    const int sw = f_row & 7;
    const int ch_hi = (f_kg ^ sw) << 4, ch_lo = ((4 + f_kg) ^ sw) << 4;
    const int rows0 = rg * 48 + f_row;
    auto opaque = [](u32x4& v) { asm volatile("" : "=v"(v)); };
    auto rd = [&](const char* ptr) -> u32x4 {
        u32x4 v;
        if (DBG & 16) opaque(v); else v = *reinterpret_cast<const u32x4*>(ptr);
        return v;
    };
    // A-step: W1 fragment nf (units 32 cg + 16 nf ..), x fragment rf (rows 48 rg + 16 rf ..)
    auto a_w = [&](int slot, int nf, int lo) { return rd(ring + slot * SLOTB + (cg * 32 + nf * 16 + f_row) * 128 + (lo ? ch_lo : ch_hi)); };
    auto a_x = [&](int slot, int rf, int lo) { return rd(ring + slot * SLOTB + X_OFF + (rows0 + rf * 16) * 128 + (lo ? ch_lo : ch_hi)); };
    // B-step: W2 fragment nf (outputs 192 half + 48 cg + 16 nf ..), G fragment rf of k-block j
    auto b_w = [&](int slot, int nf, int lo) { return rd(ring + slot * SLOTB + (cg * 48 + nf * 16 + f_row) * 128 + (lo ? ch_lo : ch_hi)); };
    auto b_g = [&](int j, int rf, int lo) { return rd(smem + OFF_G + j * G_KB + (rows0 + rf * 16) * 128 + (lo ? ch_lo : ch_hi)); };

    f32x4 acc[3][6];   // the 96 x 384 block: [row fragment][half * 3 + nf]: columns 192 half + 48 cg + 16 nf + 4 f_kg + (0..3)
    f32x4 pacc[3][2];  // P of the chunk in its A-steps
    f32x4 pold[3][2];  // P of the previous chunk, on its way through GELU
    f32x4 b1v[2];      // b1 of the chunk in pold
    u32x4 awh[2], awl[2], axh[3], axl[3];  // A-step fragments
    u32x4 bwh[3], bwl[3], bgh[3], bgl[3];  // B-step fragments

    auto load_b1 = [&](int ci) {
        const int c = chunk_of(ci < nchunks ? ci : 0);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) b1v[nf] = *reinterpret_cast<const f32x4*>(p.b1 + c * CHUNK + cg * 32 + nf * 16 + f_kg * 4);
    };
    // GELU of one accumulator fragment of pold -> (hi, lo) -> operand tile of the B-steps. Lane holds units
    // 32 cg + 16 nf + 4 f_kg + (0..3) of its rows: k-block cg of the chunk, 16-byte chunk 2 nf + (f_kg >> 1) (+ 4 for lo),
    // upper or lower 8 bytes.
    auto gelu_frag = [&](int rf, int nf) {
        char* gs = smem + OFF_G + cg * G_KB + (rows0 + rf * 16) * 128 + (f_kg & 1) * 8;
        const f32x4 v = pold[rf][nf] + b1v[nf];
        f16x4 hv, lv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = (DBG & 2) ? v[i] : gelu_erfc_as(v[i]);
            hv[i] = split_hi(g);
            lv[i] = split_lo(g, hv[i]);
        }
        const int c = 2 * nf + (f_kg >> 1);
        *reinterpret_cast<f16x4*>(gs + ((c ^ sw) << 4)) = hv;
        *reinterpret_cast<f16x4*>(gs + (((4 + c) ^ sw) << 4)) = lv;
    };

    // ================= one A-step. On entry awh / axh hold the hi fragments of this step. NEXT_B: the step after this one is
    // the first B-step (read its hi fragments instead of an A-step's).
    // GE: 0 none, 1 first half of fragment gf (nothing stored yet), 2 whole fragment gf
    auto a_step = [&](auto wait_fn, auto issue_fn, auto extra_fn, int slot, int nslot, bool next_b, int gelu_frag_idx) {
        // ---- first half: hi x hi, lo fragments in
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) awl[nf] = a_w(slot, nf, 1);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) axl[rf] = a_x(slot, rf, 1);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(awh[nf], axh[rf], pacc[rf][nf]);
        wait_fn();
        extra_fn();
        issue_fn();
        // ---- second half: the cross products; the next step's hi fragments replace the dying ones
        u32x4 nwh[3], nxh[3];
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(awl[nf], axh[rf], pacc[rf][nf]);
        if (next_b) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) nxh[rf] = b_g(0, rf, 0);
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) nwh[nf] = b_w(nslot, nf, 0);
        } else {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) nxh[rf] = a_x(nslot, rf, 0);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) nwh[nf] = a_w(nslot, nf, 0);
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) pacc[rf][nf] = mma(awh[nf], axl[rf], pacc[rf][nf]);
        if (gelu_frag_idx >= 0) gelu_frag(gelu_frag_idx >> 1, gelu_frag_idx & 1);
        if (next_b) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { bgh[i] = nxh[i]; bwh[i] = nwh[i]; }
        } else {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) axh[rf] = nxh[rf];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) awh[nf] = nwh[nf];
        }
    };

    // ================= one B-step s = 2 j + half. On entry bwh (this half's W2 hi fragments) and bgh (k-block j) are loaded;
    // bgl is loaded in the first half-step of a k-block and kept for the second.
    // next: 0 = B-step of the same k-block (new W fragments only), 1 = B-step of the next k-block, 2 = an A-step
    auto b_step = [&](auto wait_fn, auto issue_fn, auto extra_fn, int s, int slot, int nslot, int next) {
        const int half = s & 1, j = s >> 1;
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) bwl[nf] = b_w(slot, nf, 1);
        if (half == 0) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) bgl[rf] = b_g(j, rf, 1);
        }
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(bwh[nf], bgh[rf], acc[rf][half * 3 + nf]);
        wait_fn();
        extra_fn();
        issue_fn();
        u32x4 nwh[3], nxh[3];
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(bwl[nf], bgh[rf], acc[rf][half * 3 + nf]);
        if (next == 1) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) nxh[rf] = b_g(j + 1, rf, 0);
        } else if (next == 2) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) nxh[rf] = a_x(nslot, rf, 0);
        }
        if (next == 2) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) nwh[nf] = a_w(nslot, nf, 0);
        } else {
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) nwh[nf] = b_w(nslot, nf, 0);
        }
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][half * 3 + nf] = mma(bwh[nf], bgl[rf], acc[rf][half * 3 + nf]);
        if (next == 2) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) axh[rf] = nxh[rf];
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) awh[nf] = nwh[nf];
        } else {
            if (next == 1) {
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) bgh[rf] = nxh[rf];
            }
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) bwh[nf] = nwh[nf];
        }
    };

    // ---- prologue: the first four steps' DMA, then the first step's hi fragments
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) issue_step(0, t - NA);
    bool valid[3];
    int mrow[3];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
        const int m = m0 + rows0 + rf * 16;
        valid[rf] = m < p.M;
        mrow[rf] = valid[rf] ? m : p.M - 1;  // rows past M read row M - 1; nothing of them is ever stored
    }
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // step -12 has landed when at most the pieces of steps -11, -10, -9 are outstanding
    wait_and_barrier<3 * n_ops(0, 0), 3 * n_ops(0, 1)>(rg);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) awh[nf] = a_w(0, nf, 0);
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) axh[rf] = a_x(0, rf, 0);

    // ---- peeled A-steps of the first chunk; the residual rows (147 KB per workgroup, fp32) trickle in under them, two loads
    // per step in steps 0-8: plain loads into the accumulators, issued between the counted wait and the step's DMA so that they
    // are covered by the same counts (EX extra loads issued at the barrier of step t sit between the pieces of steps t + 3 and
    // t + 4 in program order: two more outstanding operations are allowed at the barriers of steps t + 1 and t + 2)
#pragma unroll
    for (int kt = 0; kt < NA; ++kt) {
        const int t = kt - NA;
        constexpr int NX = 2;
        auto wait_fn = [&]() {
            // allowed outstanding: pieces of steps t + 2, t + 3 and the extra loads issued at the barriers of steps t - 2, t - 1
            const int e = ((kt >= 1 && kt <= 9) ? NX : 0) + ((kt >= 2 && kt <= 10) ? NX : 0);
            // (all four are A-steps in this phase except past the end: steps 0, 1 of the loop are A-steps too)
            switch (e) {
                case 0: wait_and_barrier<2 * 4, 2 * 3>(rg); break;
                case NX: wait_and_barrier<2 * 4 + NX, 2 * 3 + NX>(rg); break;
                default: wait_and_barrier<2 * 4 + 2 * NX, 2 * 3 + 2 * NX>(rg); break;
            }
        };
        auto extra_fn = [&]() {
            if (kt <= 8) {
#pragma unroll
                for (int u = 0; u < NX; ++u) {
                    const int i = kt * NX + u, rf = i / 6, cf = i % 6;
                    const int n = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4;
                    acc[rf][cf] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mrow[rf] * E + n);
                }
            }
        };
        auto issue_fn = [&]() { issue_step(0, t + NSLOT); };
        a_step(wait_fn, issue_fn, extra_fn, kt & 3, (kt + 1) & 3, false, -1);
    }
    // + b2 (the residual loads were covered by the wait of the last peeled step)
#pragma unroll
    for (int cf = 0; cf < 6; ++cf) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) acc[rf][cf] += bv;
    }
    load_b1(0);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8) | (0));  // vmcnt(0): b2, b1 in; (DMA pieces too - once per launch)

    for (int it = 0; it < nchunks; ++it) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                pold[rf][nf] = pacc[rf][nf];
                pacc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        // ================= A-steps of chunk it + 1 over the GELU of chunk it: fragment f in steps 2 f, 2 f + 1 ... the writes
        // of G must be complete at the barrier of step 11 (the first B-step's hi fragments are read behind it)
#pragma unroll
        for (int kt = 0; kt < NA; ++kt) {
            auto wait_fn = [&]() {
                // pieces of steps t + 2, t + 3 (A A | A B | B B by position); the first iteration's vmcnt(0) above makes every
                // count an upper bound there
                if (kt <= 8) wait_and_barrier<n_ops(0, 0) * 2, n_ops(0, 1) * 2>(rg);
                else if (kt == 9) wait_and_barrier<n_ops(0, 0) + 3, n_ops(0, 1) + 3>(rg);
                else wait_and_barrier<6, 6>(rg);
            };
            auto issue_fn = [&]() { issue_step(it, kt + NSLOT); };
            auto extra_fn = [&]() {};
            // GELU schedule: six fragments over steps 0..10 (two steps per fragment would need half-fragment state; one
            // fragment every other step keeps the code simple: steps 0, 2, 4, 6, 8, 10)
            const int gf = (kt % 2 == 0 && kt <= 10) ? kt / 2 : -1;
            a_step(wait_fn, issue_fn, extra_fn, kt & 3, (kt + 1) & 3, kt == NA - 1, gf);
        }
        // ================= B-steps of chunk it
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int t = NA + s;
            auto wait_fn = [&]() {
                // pieces of steps t + 2, t + 3: B B up to t = 16, B A' at 17, A' A' at 18, 19; + the two b1 loads issued at
                // the barrier of step 12 (allowed outstanding at the barriers of steps 13, 14)
                if (t == 13 || t == 14) wait_and_barrier<6 + 2, 6 + 2>(rg);
                else if (t <= 16) wait_and_barrier<6, 6>(rg);
                else if (t == 17) wait_and_barrier<3 + n_ops(0, 0), 3 + n_ops(0, 1)>(rg);
                else wait_and_barrier<2 * n_ops(0, 0), 2 * n_ops(0, 1)>(rg);
            };
            auto issue_fn = [&]() { issue_step(it, t + NSLOT); };
            auto extra_fn = [&]() { if (s == 0) load_b1(it + 1); };
            b_step(wait_fn, issue_fn, extra_fn, s, t & 3, (t + 1) & 3, s == NB - 1 ? 2 : (s & 1));
        }
    }

    // ---- LayerNorm epilogue: G is out of use, its region carries the statistics exchange; the ring may still receive the
    // out-of-bounds fillers issued past the last chunk
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | (0));  // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    float* stat = reinterpret_cast<float*>(smem + OFF_G);
    float mean[3], rstd[3];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
        float sm = 0.f;
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const f32x4 v = acc[rf][cf];
            sm += (v[0] + v[1]) + (v[2] + v[3]);
        }
        sm += __shfl_xor(sm, 16);
        sm += __shfl_xor(sm, 32);
        if (f_kg == 0) stat[cg * BM + rows0 + rf * 16] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
        const int r = rows0 + rf * 16;
        mean[rf] = ((stat[r] + stat[BM + r]) + (stat[2 * BM + r] + stat[3 * BM + r])) * (1.0f / E);
        float q = 0.f;
#pragma unroll
        for (int cf = 0; cf < 6; ++cf)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = acc[rf][cf][k] - mean[rf];
                q = __builtin_fmaf(d, d, q);
            }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (f_kg == 0) stat[(4 + cg) * BM + r] = q;
    }
    __syncthreads();
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
        const int r = rows0 + rf * 16;
        const float var = ((stat[4 * BM + r] + stat[5 * BM + r]) + (stat[6 * BM + r] + stat[7 * BM + r])) * (1.0f / E);
        rstd[rf] = 1.0f / sqrtf(var + p.eps);
    }
#pragma unroll
    for (int cf = 0; cf < 6; ++cf) {
        const int n = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + n), b = *reinterpret_cast<const f32x4*>(p.beta + n);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            if (!valid[rf]) continue;
            const size_t off = (size_t)(m0 + rows0 + rf * 16) * E + n;
            const f32x4 v = acc[rf][cf];
            const float mu = mean[rf], rs = rstd[rf];
            const f32x4 hv = {(v[0] - mu) * rs * g[0] + b[0], (v[1] - mu) * rs * g[1] + b[1], (v[2] - mu) * rs * g[2] + b[2],
                              (v[3] - mu) * rs * g[3] + b[3]};
            *reinterpret_cast<f32x4*>(p.x_out + off) = v;
            split_store4(p.h_out, off, hv);
        }
    }
}

// Packs W1 (F, 384) and W2 (384, F), both split row-major, into the stream the kernel consumes: per hidden chunk c of 128
// units 12 blocks [128 units][128 B] (k-block kb of W1 rows 128 c ..) followed by 8 blocks [192 outputs][128 B] (block
// s = 2 j + half: W2 rows 192 half .., k-block 4 c + j), every 128-byte line stored with its 16-byte chunks XOR-swizzled by
// (line & 7) - the LDS image. One thread per 16-byte chunk.
__global__ void pack_kernel(const char* __restrict__ w1, const char* __restrict__ w2, char* __restrict__ out, int F) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte chunk of the output
    const long long total = (long long)(F / CHUNK) * (CHUNK_BYTES / 16);
    if (idx >= total) return;
    const int c = (int)(idx / (CHUNK_BYTES / 16));
    const int r = (int)(idx % (CHUNK_BYTES / 16));
    const char* src;
    if (r < B_PART / 16) {
        const int kb = r / (A_BLOCK / 16), q = r % (A_BLOCK / 16), line = q >> 3, pc = q & 7;
        src = w1 + ((size_t)(c * CHUNK + line) * E + kb * 32) * 4 + ((pc ^ (line & 7)) << 4);
    } else {
        const int rb = r - B_PART / 16;
        const int s = rb / (B_BLOCK / 16), q = rb % (B_BLOCK / 16), line = q >> 3, pc = q & 7;
        const int n = (s & 1) * (E / 2) + line, kblk = c * 4 + (s >> 1);
        src = w2 + ((size_t)n * F + kblk * 32) * 4 + ((pc ^ (line & 7)) << 4);
    }
    *reinterpret_cast<u32x4*>(out + idx * 16) = *reinterpret_cast<const u32x4*>(src);
}

}  // namespace ffs
}  // namespace pp

extern "C" long long pp_ffn_split_packed_bytes(int E, int F) {
    using namespace pp;
    if (E != ffs::E || F <= 0 || F % ffs::CHUNK != 0) return -1;
    return (long long)(F / ffs::CHUNK) * ffs::CHUNK_BYTES;
}

extern "C" int pp_ffn_split_pack_weights(const void* w1, const void* w2, void* packed, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(w1 && w2 && packed, PP_ERR_INVALID_ARG, "pp_ffn_split_pack_weights: NULL argument");
    PP_REQUIRE(E == ffs::E && F > 0 && F % ffs::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_ffn_split_pack_weights: built for embed dim 384 (ViT-S), hidden width a multiple of 128");
    const long long total = (long long)(F / ffs::CHUNK) * (ffs::CHUNK_BYTES / 16);
    hipLaunchKernelGGL(ffs::pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const char*>(w1), reinterpret_cast<const char*>(w2), reinterpret_cast<char*>(packed), F);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_ffn_split_residual_layernorm(const void* h_in, const void* w_packed, const float* b1, const float* b2,
                                               const float* residual, float* x_out, const float* gamma, const float* beta,
                                               float eps, void* h_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(h_in && w_packed && b1 && b2 && residual && x_out && gamma && beta && h_out, PP_ERR_INVALID_ARG,
               "pp_ffn_split_residual_layernorm: NULL argument");
    PP_REQUIRE(E == ffs::E, PP_ERR_UNSUPPORTED, "pp_ffn_split_residual_layernorm: built for embed dim 384 (ViT-S)");
    PP_REQUIRE(M > 0 && F > 0 && F % ffs::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_ffn_split_residual_layernorm: hidden width must be a positive multiple of 128");
    PP_REQUIRE((size_t)M * E * 4 < ffs::OOB && (size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES < ffs::OOB, PP_ERR_UNSUPPORTED,
               "pp_ffn_split_residual_layernorm: operand exceeds 2 GiB");
    ffs::Params p{};
    p.h = h_in;
    p.wpack = w_packed;
    p.b1 = b1;
    p.b2 = b2;
    p.residual = residual;
    p.x_out = x_out;
    p.gamma = gamma;
    p.beta = beta;
    p.h_out = h_out;
    p.M = M;
    p.F = F;
    p.h_bytes = (unsigned)((size_t)M * E * 4);
    p.w_bytes = (unsigned)((size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES);
    p.eps = eps;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffs::ffn_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ffs::LDS));
    hipLaunchKernelGGL(ffs::ffn_split_kernel, dim3((M + ffs::BM - 1) / ffs::BM), dim3(ffs::THREADS), ffs::LDS,
                       reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
