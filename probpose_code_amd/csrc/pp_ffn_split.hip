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
//     wave (rg, cg): rows 48 rg .. +47, column quarter cg; waves w and w + 4 share a SIMD;
//   * the hidden layer runs in chunks of 128 units; per chunk twenty steps on a ring of FOUR 28 KiB slots:
//       A-step kb (12 per chunk)   P += x[:, kb] W1[chunk, kb]^T   slot = W1 block (128 lines x 128 B) + x block (96 lines);
//                                  wave tile 48 rows x 32 units, 18 MFMAs
//       B-step (j, half) (8)       acc[:, half] += G[:, j] W2[half, chunk j]^T   slot = W2 half block (192 lines);
//                                  wave tile 48 rows x 48 outputs, 27 MFMAs; the G fragments stay for both halves
//     software-pipelined across chunks like pp_mlp.hip: the loop body is [A-steps of chunk c + 1 | B-steps of chunk c];
//   * a step of a wave is two segments, L (its DMA share of step s + 3, every fragment read of step s, the side work)
//     and C (the MFMAs, nothing else), and the two waves of a SIMD are never in the same kind: waves 4-7 run one segment
//     behind waves 0-3, one barrier per segment (see the main loop);
//   * GELU(P + b1) of a finished chunk (fp32 -> erfc form -> (hi, lo)) rides in the L segments of the NEXT twenty steps: one
//     value pair per B-step (held in registers: the G tile is still being read), the last four pairs and the stores into
//     the G tile (48 KiB, operand of the B-steps) in the following A-steps;
//   * odd chunk visits walk the k-blocks of their A-steps backwards (FFS_SAWTOOTH): the rows a workgroup re-streams per chunk are
//     then found in L2 where the previous pass left off - cyclic re-reads of a working set larger than the cache never hit
//     (HBM-side traffic 730 -> 456 MB per launch);
//   * W1 / W2 come PRE-PACKED in consumption order (pp_ffn_split_pack_weights: per chunk 12 W1 blocks of 16 KiB, then
//     8 W2 half blocks of 24 KiB, 128-byte lines with the LDS XOR swizzle already applied), so a weight DMA instruction
//     is a linear 1 KiB copy; the x lines are 128-byte segments of the row-major split tensor, swizzled at the source;
//   * the 96 x 384 accumulators start from residual + b2 (requested ahead of the first DMA piece - never between pieces: plain
//     loads and LDS-DMA pieces do not retire in order with respect to each other, see the projection phase) and end in the
//     LayerNorm epilogue (row statistics in registers, one LDS exchange between the column quarters).
// LDS: 48 KiB G + 4 x 28 KiB ring = 160 KiB.
//
// Measured at bs 64 (M = 24 576, F = 1536; scripts/micro/ffs_variants.sh + ffs_variants_bench.py, round-robin minima):
// 176 - 180 us against 232 us for pp_linear_ovl (fc1 + GELU) + pp_gemm_ln (fc2 + residual + LayerNorm), and 302 MB of HBM
// traffic less per layer. Matrix pipe 48 % busy. What the ablations said (FFS_DBG / FFS_XSRC, same instruction stream):
// MFMAs alone 110 us (the pipe's own time: 5400 per wave x 16 cycles + 25 us of prologue / epilogue), everything but the
// MFMAs 116 us, no DMA traffic (empty descriptors) 155 us. Chunk order: every workgroup of an XCD walks the chunks in the SAME
// order (rotated per XCD only) - W1 / W2 are 4.5 MiB, more than the 4 MiB L2, and with a rotation per workgroup (what
// pp_mlp.hip does with its 3.2 MB) every chunk was in flight somewhere all the time: fill-only skeleton 143 -> 107 us.
// All eight waves in the same phase (the form of pp_mlp.hip: next step's hi fragments read under the cross products,
// DMA pieces and GELU placed between the MFMAs with sched_group_barrier) measured the same 180 us.
#include "pp_common.h"
#include "pp_split.h"
#include "pp_ffn_params.h"

namespace pp {
namespace ffs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef FFS_DBG
#define FFS_DBG 0  // dev ablations (timing only, wrong results): 2 no GELU, 4 no MFMA, 8 no DMA, 16 no fragment reads; 512 time stamps (pp_ffs_set_trace)
#endif
constexpr int DBG = FFS_DBG;

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


__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (DBG & 4) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// DMA instructions of one step by wave half: A-steps 16 W1 pieces by waves 0-3 (4 each) + 12 x pieces by waves 4-7 (3 each),
// B-steps 24 pieces, 3 per wave
__host__ __device__ constexpr bool is_a(int t) { return ((t % STEPS) + STEPS) % STEPS < NA; }
#ifndef FFS_ONEHALF
#define FFS_ONEHALF 0  // dev A/B: 1 = waves 0-3 issue ALL DMA pieces (7 per A-step, 6 per B-step), waves 4-7 none
#endif
__host__ __device__ constexpr int n_ops(int t, int rg) { return FFS_ONEHALF ? (rg == 0 ? (is_a(t) ? 7 : 6) : 0) : (is_a(t) ? (rg == 0 ? 4 : 3) : 3); }

#define FFS_WAIT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (0 << 8) | (((N) >> 4) << 14))  // vmcnt(N) lgkmcnt(0)

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

// (the body is a __device__ function template and the two kernels below plain functions: a KERNEL template with
// value-returning lambdas inside loses its host stub in the host pass)
template <bool PROJ>
__device__ __forceinline__ void ffn_split_body(const Params p) {  // (by value: behind a reference `rg == 0 ? p.wpack : p.h` became an indexed scratch load)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nchunks = p.F / CHUNK;
    // the workgroups of an XCD walk the hidden chunks in the same order, the XCDs in different rotations
#ifndef FFS_ROT
#define FFS_ROT 2  // dev A/B switch: 0 no rotation, 1 by rank inside the XCD, 2 by XCD (see the file header)
#endif
    const int c_rot = FFS_ROT == 1 ? (int)(blockIdx.x >> 3) % nchunks : FFS_ROT == 2 ? (int)(blockIdx.x & 7) % nchunks : 0;
    auto chunk_of = [&](int i) { const int c = i + c_rot; return c >= nchunks ? c - nchunks : c; };

    char* const ring = smem + OFF_RING;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (!(DBG & 512)) return;
        if (blockIdx.x != 0 || (wv & 3) != 0) return;
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0 && stamp_i < 2048) p.trace[(wv >> 2) * 2048 + stamp_i] = t;
        ++stamp_i;
    };

    // ---- DMA addressing. A weight piece is a linear 1 KiB copy (lane i -> byte 16 i). An x piece is 8 rows x 128 B: lane
    // (row l = lane >> 3, physical chunk pc = lane & 7) fetches logical chunk pc ^ l (rows 8 p + l: (row & 7) == l).
    // Every wave issues THREE pieces per step with the same instructions - what differs by wave half is data (descriptor,
    // offsets, LDS destination), selected once - so that the pieces can sit between the MFMAs of a half-step as one basic
    // block: a buffer_load ... lds holds its wave for 60 - 180 cycles when the texture path is busy (16 cycles per KiB,
    // 28 KiB per A-step), and issued as a burst behind the barrier all eight waves - and the matrix pipe - waited for it.
    //   A-step: waves 0-3 W1 pieces 3 w .. 3 w + 2 (+ one extra piece 12 + w, the only wave-half-dependent instruction),
    //           waves 4-7 x pieces 3 (w - 4) .. + 2;      B-step: W2 pieces 3 w .. 3 w + 2.
    const int x_l = lane >> 3;
    const unsigned v_w = (unsigned)lane * 16u;
    const unsigned v_x = (unsigned)(m0 + 24 * (wv & 3) + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
    const unsigned v_a = rg == 0 ? v_w : v_x;                       // voffset of an A-step piece
    const int a_stride = rg == 0 ? 1024 : 8 * E * 4;                // soffset step from piece to piece
    const int a_dst = rg == 0 ? 3 * wv * 1024 : X_OFF + 3 * (wv & 3) * 1024;
#ifndef FFS_XSRC
#define FFS_XSRC 0  // dev (timing only): 2 no x traffic, 3 no DMA traffic at all (every descriptor empty: same instructions, zeros)
#endif
    auto rsrc_a = [&](bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rg == 0 ? p.wpack : p.h), 0,
                                                 (live && FFS_XSRC != 3) ? (rg == 0 ? p.w_bytes : (FFS_XSRC == 2 ? 0u : p.h_bytes)) : 0u, 0x00020000);
    };
    auto rsrc_w = [&](bool live) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpack), 0, (live && FFS_XSRC != 3) ? p.w_bytes : 0u, 0x00020000); };
    // piece u (0..2; 3 = the extra W1 piece of waves 0-3) of A-step kb of the chunk visited ci-th into ring slot `slot`; past the
    // last chunk the descriptor has no extent: the DMA writes zeros, every wave's vmcnt arithmetic stays the same
#ifndef FFS_SAWTOOTH
#define FFS_SAWTOOTH 1  // 0 = every chunk walks the k-blocks 0 .. 11 (dev A/B)
#endif
    auto issue_a = [&](int ci, int kb_, int slot, int u) {
        if (DBG & 8) return;
        const bool live = ci < nchunks;
        // Odd visits walk the k-blocks backwards. A workgroup re-reads its 144 KB of rows once per chunk, the 32 workgroups of an
        // XCD hold 4.6 MB of them beside the weights: more than the 4 MB L2, and walked in the same direction every time an LRU
        // cache never hits. Turning round at the end of every pass finds the blocks read last still resident.
        const int kb = (FFS_SAWTOOTH && (ci & 1)) ? NA - 1 - kb_ : kb_;
        const int blk = chunk_of(live ? ci : 0) * CHUNK_BYTES + kb * A_BLOCK;
        if (FFS_ONEHALF) {  // wave w < 4: W1 pieces 4 w .. 4 w + 3 (u < 4), x pieces 3 w .. 3 w + 2 (u = 4 .. 6)
            if (rg != 0 || u > 6) return;
            if (u < 4) {
                const int q = 4 * wv + u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w(live), (lds_ptr_t)(ring + slot * SLOTB + q * 1024), 16, v_w, blk + q * 1024, 0, 0);
            } else {
                const int xq = 3 * wv + (u - 4);
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.h), 0, (live && FFS_XSRC != 3 && FFS_XSRC != 2) ? p.h_bytes : 0u, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(ring + slot * SLOTB + X_OFF + xq * 1024), 16, v_x, kb * 128 + (u - 4) * 8 * E * 4, 0, 0);
            }
            return;
        }
        if (u < 3) {
            const int so = (rg == 0 ? blk + 3 * wv * 1024 : kb * 128) + u * a_stride;
#ifndef FFS_X_AUX
#define FFS_X_AUX 0  // dev: cache policy bits of the x pieces (2 = nt: the rows are private to the workgroup, no reuse in L2 to protect)
#endif
            if (FFS_X_AUX != 0 && rg == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a(live), (lds_ptr_t)(ring + slot * SLOTB + a_dst + u * 1024), 16, v_a, so, 0, FFS_X_AUX);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a(live), (lds_ptr_t)(ring + slot * SLOTB + a_dst + u * 1024), 16, v_a, so, 0, 0);
        } else if (rg == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w(live), (lds_ptr_t)(ring + slot * SLOTB + (12 + wv) * 1024), 16, v_w, blk + (12 + wv) * 1024, 0, 0);
        }
    };
    auto issue_b = [&](int ci, int sb, int slot, int u) {
        if (DBG & 8) return;
        const bool live = ci >= 0 && ci < nchunks;
        if (FFS_ONEHALF) {  // wave w < 4: W2 pieces 6 w .. 6 w + 5
            if (rg != 0 || u > 5) return;
            const int q = 6 * wv + u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w(live), (lds_ptr_t)(ring + slot * SLOTB + q * 1024), 16, v_w,
                                                     chunk_of(live ? ci : 0) * CHUNK_BYTES + B_PART + sb * B_BLOCK + q * 1024, 0, 0);
            return;
        }
        const int so = chunk_of(live ? ci : 0) * CHUNK_BYTES + B_PART + sb * B_BLOCK + 3 * wv * 1024;
        if (u < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w(live), (lds_ptr_t)(ring + slot * SLOTB + (3 * wv + u) * 1024), 16, v_w, so + u * 1024, 0, 0);
    };
    // step index t relative to the start of iteration `it` (which runs A-steps of chunk it + 1 and B-steps of chunk it);
    // t < 0: the peeled A-steps of chunk 0 (t = -12 .. -1), t >= 20: the next iteration
    auto issue_piece = [&](int it, int t, int u) {
        const int slot = (t + 4 * STEPS) & (NSLOT - 1);
        if (t < 0) issue_a(0, t + NA, slot, u);
        else if (t < NA) issue_a(it + 1, t, slot, u);
        else if (t < STEPS) issue_b(it, t - NA, slot, u);
        else issue_a(it + 2, t - STEPS, slot, u);
    };
    auto issue_step = [&](int it, int t) {
#pragma unroll
        for (int u = 0; u < (FFS_ONEHALF ? 7 : 4); ++u) issue_piece(it, t, u);
    };

    // ---- fragment reads: hi halves in 16-byte chunk f_kg, lo halves in chunk 4 + f_kg of a line, swizzled by line & 7.
    // Six per-lane byte offsets (three line sets x hi / lo) and a slot offset the compiler cannot fold (so that it does not keep
    // a hoisted address register per (slot, line set): the loop has none to spare); fragment strides are instruction offsets.
    const int sw = f_row & 7;
    const int ch_hi = (f_kg ^ sw) << 4, ch_lo = ((4 + f_kg) ^ sw) << 4;
    const int rows0 = rg * 48 + f_row;
    const int la[2] = {OFF_RING + (cg * 32 + f_row) * 128 + ch_hi, OFF_RING + (cg * 32 + f_row) * 128 + ch_lo};  // A-step W1 lines
    const int lb[2] = {OFF_RING + (cg * 48 + f_row) * 128 + ch_hi, OFF_RING + (cg * 48 + f_row) * 128 + ch_lo};  // B-step W2 lines
    const int lx[2] = {rows0 * 128 + ch_hi, rows0 * 128 + ch_lo};                                                  // row lines (x, G)
    const int lxa[2] = {lx[0] + OFF_RING + X_OFF, lx[1] + OFF_RING + X_OFF};  // x lines of an A slot (instruction offsets stay < 64 KiB)
    auto slot_off = [](int slot) { int so = slot * SLOTB; asm volatile("" : "+s"(so)); return so; };
    auto opaque = [](u32x4& v) { asm volatile("" : "=v"(v)); };
    auto rd = [&](int off) -> u32x4 {
        u32x4 v;
        if (DBG & 16) opaque(v); else v = *reinterpret_cast<const u32x4*>(smem + off);
        return v;
    };
    // A-step: W1 fragment nf (units 32 cg + 16 nf ..), x fragment rf (rows 48 rg + 16 rf ..); so = slot_off(slot)
    auto a_w = [&](int so, int nf, int lo) { return rd(la[lo] + so + nf * 2048); };
    auto a_x = [&](int so, int rf, int lo) { return rd(lxa[lo] + so + rf * 2048); };
    // B-step: W2 fragment nf (outputs 192 half + 48 cg + 16 nf ..), G fragment rf of k-block j
    auto b_w = [&](int so, int nf, int lo) { return rd(lb[lo] + so + nf * 2048); };
    auto b_g = [&](int j, int rf, int lo) { return rd(lx[lo] + OFF_G + j * G_KB + rf * 2048); };

    f32x4 acc[3][6];   // the 96 x 384 block: [row fragment][half * 3 + nf]: columns 192 half + 48 cg + 16 nf + 4 f_kg + (0..3)
    f32x4 pacc[3][2];  // P of the chunk in its A-steps
    f32x4 b1v[2];      // b1 of the chunk whose A-steps come next (its accumulators start from it)
    u32x4 awh[2], awl[2], axh[3], axl[3];  // A-step fragments
    u32x4 bwh[3], bwl[3], bgh[3], bgl[3];  // B-step fragments

    auto load_b1 = [&](int ci) {
        const int c = chunk_of(ci < nchunks ? ci : 0);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) b1v[nf] = *reinterpret_cast<const f32x4*>(p.b1 + c * CHUNK + cg * 32 + nf * 16 + f_kg * 4);
    };
    // GELU of the chunk whose A-steps have just ended runs under the B-steps of the chunk BEFORE it (one value pair per step; the last four pairs under
    // the first A-steps of the next iteration, from a copy):
    // the A-steps carry the fragment reads of two streamed operands and the x DMA, the B-steps have the issue slots to spare.
    // The G tile is still being read then, so the (hi, lo) pairs wait in registers (24) and are stored at the head of the next
    // iteration. Lane holds units 32 cg + 16 nf + 4 f_kg + (0..3) of its rows: k-block cg of the chunk, 16-byte chunk
    // 2 nf + (f_kg >> 1) (+ 4 for lo), upper or lower 8 bytes.
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x2 gq_hi[12], gq_lo[12];  // value pairs (2 q, 2 q + 1) of the 24 values of a lane: fragment q >> 1, elements 2 (q & 1) ..
    // (written stage by stage over all values of the call: 2 * count independent dependency chains in program order, so that
    // consecutive VALU instructions between two MFMAs do not wait for each other - a single chain issues one instruction per
    // ~8 cycles and the matrix pipe idles behind it)
    f32x4 pold[2];  // fragments 4, 5 of the finished chunk: their GELU runs under the first A-steps of the next iteration
    auto gelu_pairs = [&](int first, int count, bool from_pold) {
        constexpr int MAXV = 4;
        float x[MAXV], z[MAXV], t[MAXV], q[MAXV], e[MAXV], g[MAXV];
        const int n = 2 * count;
#pragma unroll
        for (int u = 0; u < n; ++u) {
            const int qq = first + (u >> 1), f = qq >> 1, i = 2 * (qq & 1) + (u & 1);
            x[u] = from_pold ? pold[f - 4][i] : pacc[f >> 1][f & 1][i];  // (b1 is already in: the chunk's accumulators start from it)
        }
        if (DBG & 2) {
#pragma unroll
            for (int u = 0; u < n; ++u) g[u] = x[u];
        } else {
            // gelu_erfc_as of pp_split.h (Abramowitz & Stegun 7.1.26), same operations in the same order per value
#pragma unroll
            for (int u = 0; u < n; ++u) z[u] = fabsf(x[u]) * 0.70710678118654752440f;
#pragma unroll
            for (int u = 0; u < n; ++u) t[u] = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z[u], 1.0f));
#pragma unroll
            for (int u = 0; u < n; ++u) e[u] = __builtin_amdgcn_exp2f(-(z[u] * z[u]) * 1.44269504088896340736f);
#pragma unroll
            for (int u = 0; u < n; ++u) q[u] = __builtin_fmaf(t[u], 1.061405429f, -1.453152027f);
#pragma unroll
            for (int u = 0; u < n; ++u) q[u] = __builtin_fmaf(t[u], q[u], 1.421413741f);
#pragma unroll
            for (int u = 0; u < n; ++u) q[u] = __builtin_fmaf(t[u], q[u], -0.284496736f);
#pragma unroll
            for (int u = 0; u < n; ++u) q[u] = __builtin_fmaf(t[u], q[u], 0.254829592f);
#pragma unroll
            for (int u = 0; u < n; ++u) {
                const float erfc_z = t[u] * q[u] * e[u];
                g[u] = 0.5f * x[u] * (x[u] < 0.f ? erfc_z : 2.0f - erfc_z);
                split_pin(g[u]);  // (pp_split.h)
            }
        }
#pragma unroll
        for (int c = 0; c < count; ++c) {
            const f16x2 h = {split_hi(g[2 * c]), split_hi(g[2 * c + 1])};
            const f16x2 l = {split_lo(g[2 * c], h[0]), split_lo(g[2 * c + 1], h[1])};
            gq_hi[first + c] = h;
            gq_lo[first + c] = l;
            // pinned here: nothing reads the pairs before the next iteration, and the compiler would sink the arithmetic there
            asm volatile("" : "+v"(gq_hi[first + c]), "+v"(gq_lo[first + c]));
        }
    };
#ifndef FFS_GELU_IN_C
#define FFS_GELU_IN_C 0  // 1 = the B-steps' GELU pair runs in the MFMA segment C(s), two or three VALU instructions behind each MFMA (gelu_stage), not in L(s): measured 200.0 vs 198.7 us, same bits. (In that build the compiler sinks the stages to their first use - one clump behind the last MFMAs; with every stage's values pinned by an empty asm the ISA does alternate MFMA / two VALU as intended and measures 198.8 vs 197.6 us: the erfc arithmetic costs its ~17 us per launch wherever it is issued - both waves of a SIMD are short of issue time, not one of them)
#endif
    // The same pair, cut into 24 stages of two or three VALU instructions (a transcendental has a stage to itself): stage k sits
    // behind the k-th MFMA of a B-step's C segment - a 16x16x32 MFMA holds the matrix pipe for 16 cycles and takes 4 to issue,
    // which leaves the wave three plain VALU issue slots per MFMA at no cost to the pipe. Same operations in the same order per
    // value as gelu_pairs.
    float sx[2], sz[2], st[2], se[2], sq[2], sg[2];
    _Float16 sh[2], sl[2];
    auto gelu_stage = [&](int k, int pair) {
        const int f = pair >> 1, i0 = 2 * (pair & 1);
        if (DBG & 2) {
            if (k == 0) {
                sg[0] = pacc[f >> 1][f & 1][i0]; sg[1] = pacc[f >> 1][f & 1][i0 + 1];
            }
        } else switch (k) {
            case 0: sx[0] = pacc[f >> 1][f & 1][i0]; sx[1] = pacc[f >> 1][f & 1][i0 + 1];
                    sz[0] = fabsf(sx[0]) * 0.70710678118654752440f; sz[1] = fabsf(sx[1]) * 0.70710678118654752440f; break;
            case 1: st[0] = __builtin_fmaf(0.3275911f, sz[0], 1.0f); st[1] = __builtin_fmaf(0.3275911f, sz[1], 1.0f); break;
            case 2: st[0] = __builtin_amdgcn_rcpf(st[0]); break;
            case 3: st[1] = __builtin_amdgcn_rcpf(st[1]); break;
            case 4: se[0] = sz[0] * sz[0]; se[1] = sz[1] * sz[1]; break;
            case 5: se[0] = -se[0] * 1.44269504088896340736f; se[1] = -se[1] * 1.44269504088896340736f; break;
            case 6: se[0] = __builtin_amdgcn_exp2f(se[0]); break;
            case 7: se[1] = __builtin_amdgcn_exp2f(se[1]); break;
            case 8: sq[0] = __builtin_fmaf(st[0], 1.061405429f, -1.453152027f); sq[1] = __builtin_fmaf(st[1], 1.061405429f, -1.453152027f); break;
            case 9: sq[0] = __builtin_fmaf(st[0], sq[0], 1.421413741f); sq[1] = __builtin_fmaf(st[1], sq[1], 1.421413741f); break;
            case 10: sq[0] = __builtin_fmaf(st[0], sq[0], -0.284496736f); sq[1] = __builtin_fmaf(st[1], sq[1], -0.284496736f); break;
            case 11: sq[0] = __builtin_fmaf(st[0], sq[0], 0.254829592f); sq[1] = __builtin_fmaf(st[1], sq[1], 0.254829592f); break;
            case 12: sq[0] = st[0] * sq[0]; sq[1] = st[1] * sq[1]; break;
            case 13: sq[0] = sq[0] * se[0]; sq[1] = sq[1] * se[1]; break;                       // erfc(z)
            case 14: se[0] = 2.0f - sq[0]; se[1] = 2.0f - sq[1]; break;
            case 15: sq[0] = sx[0] < 0.f ? sq[0] : se[0]; break;
            case 16: sq[1] = sx[1] < 0.f ? sq[1] : se[1]; break;
            case 17: sg[0] = 0.5f * sx[0]; sg[1] = 0.5f * sx[1]; break;
            case 18: sg[0] = sg[0] * sq[0]; sg[1] = sg[1] * sq[1]; split_pin(sg[0]); split_pin(sg[1]); break;
            default: break;
        }
        switch (k) {
            case 19: sh[0] = split_hi(sg[0]); sh[1] = split_hi(sg[1]); break;
            case 20: sz[0] = (float)sh[0]; sz[1] = (float)sh[1]; break;
            case 21: sz[0] = sg[0] - sz[0]; sz[1] = sg[1] - sz[1]; break;
            case 22: sl[0] = (_Float16)sz[0]; sl[1] = (_Float16)sz[1]; break;
            case 23: {
                gq_hi[pair] = f16x2{sh[0], sh[1]};
                gq_lo[pair] = f16x2{sl[0], sl[1]};
                asm volatile("" : "+v"(gq_hi[pair]), "+v"(gq_lo[pair]));
                break;
            }
            default: break;
        }
    };
    auto write_g = [&](int f0, int f1) {
#pragma unroll
        for (int f = f0; f < f1; ++f) {
            const int rf = f >> 1, nf = f & 1;
            char* gs = smem + OFF_G + cg * G_KB + (rows0 + rf * 16) * 128 + (f_kg & 1) * 8;
            const int c = 2 * nf + (f_kg >> 1);
            const f16x4 hv = {gq_hi[2 * f][0], gq_hi[2 * f][1], gq_hi[2 * f + 1][0], gq_hi[2 * f + 1][1]};
            const f16x4 lv = {gq_lo[2 * f][0], gq_lo[2 * f][1], gq_lo[2 * f + 1][0], gq_lo[2 * f + 1][1]};
            *reinterpret_cast<f16x4*>(gs + ((c ^ sw) << 4)) = hv;
            *reinterpret_cast<f16x4*>(gs + (((4 + c) ^ sw) << 4)) = lv;
        }
    };
    // ================= main loop, role-alternating form. The two waves of a SIMD (wave w of rows 0-47 and wave w + 4 of rows
    // 48-95) are never in the same kind of segment: a step of a wave is  L(s): its DMA share of step s + 3, all fragment reads
    // of step s, the GELU / bias / residual side work  |  C(s): the 18 / 27 MFMAs of the step, nothing else.  Waves 4-7 run one
    // segment behind waves 0-3 (one extra barrier at the start), so while one half's MFMAs own the matrix pipe the other half's
    // LDS reads, DMA issue (60 - 180 cycles per instruction when the texture path is busy) and VALU work proceed beside them -
    // with all eight waves in the same phase those costs add to the MFMA time.
    // Ring: step s is read in time slots 2 s - 1 (waves 0-3) and 2 s (waves 4-7); L(s) refills the slot of step s - 1 with step
    // s + 3; the wave half that ends an even time slot (0-3 after C(s), 4-7 after L(s)) waits for its pieces of step s + 1.
    auto sync_l = [&](auto wait_rg1) {  // after L(s)
        __builtin_amdgcn_sched_barrier(0);
        stamp();
#ifndef FFS_RG0_NOWAIT
#define FFS_RG0_NOWAIT 1  // 1 (shipped) = waves 0-3 do not wait for their fragment reads at the end of L(s): their slot is refilled two time slots later, and the MFMAs of C(s) get the compiler's own partial waits (211.1 -> 207.1 us, same bits); 0 = lgkmcnt(0) there
#endif
        if (rg == 0) { if (!FFS_RG0_NOWAIT) __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14)); }  // lgkmcnt(0): the fragments are in
        else wait_rg1();
        __builtin_amdgcn_s_barrier();
        stamp();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_c = [&](auto wait_rg0) {  // after C(s)
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        if (rg == 0) wait_rg0();
        __builtin_amdgcn_s_barrier();
        stamp();
        __builtin_amdgcn_sched_barrier(0);
    };
    // allowed outstanding at the wait for step S + 1: pieces of steps S + 2, S + 3 (own share) + the extra loads of L(S - 1), L(S)
    // L segments: this wave's DMA pieces alternate with its fragment reads - four waves issue at the same time, the texture
    // path takes one piece per 16 cycles, and whoever finds its queue full stands still: the reads go out in those gaps
#ifndef FFS_DMA_IN_C
#define FFS_DMA_IN_C 2  // 0 = all DMA pieces of step s + 3 in the load segment L(s); 1 = the DMA pieces of step s + 3 are issued in the MFMA segment C(s), between the MFMAs, not in L(s); 2 (shipped) = two in L(s), the rest in C(s): 181.6 -> 179.4 us (FFN), 215.2 -> 211.9 (with projection)
#endif
    auto pin = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto a_load = [&](int slot_i, auto issue_fn_) {
        auto issue_fn = [&](int u) {
            if (FFS_ONEHALF) { issue_fn_(2 * u); if (u < 3) issue_fn_(2 * u + 1); return; }  // 0 1 | 2 3 | 4 5 | 6
            if (FFS_DMA_IN_C == 0 || (FFS_DMA_IN_C == 2 && u < 2)) issue_fn_(u);
        };
        const int so = slot_off(slot_i);
        issue_fn(0);
        pin();
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) { awh[nf] = a_w(so, nf, 0); awl[nf] = a_w(so, nf, 1); }
        pin();
        issue_fn(1);
        pin();
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) axh[rf] = a_x(so, rf, 0);
        pin();
        issue_fn(2);
        pin();
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) axl[rf] = a_x(so, rf, 1);
        pin();
        issue_fn(3);
    };
#ifndef FFS_CPRIO
#define FFS_CPRIO 0  // dev A/B: n > 0 = a wave raises its priority to n for its MFMA segments C(s) (the partner wave of the SIMD is in a load segment then): 207.9 / 207.6 us (n = 1 / 3) against 205.8 - the matrix pipe does not wait for issue slots
#endif
    auto a_compute = [&](auto issue_fn_) {
        auto issue_fn = [&](int u) { if (!FFS_ONEHALF && (FFS_DMA_IN_C == 1 || (FFS_DMA_IN_C == 2 && u >= 2))) { pin(); issue_fn_(u); pin(); } };
        if (FFS_CPRIO) __builtin_amdgcn_s_setprio(FFS_CPRIO);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(awh[nf], axh[rf], pacc[rf][nf]);
            if (rf == 1) issue_fn(0);
        }
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(awl[nf], axh[rf], pacc[rf][nf]);
            if (rf == 0) issue_fn(1);
            if (rf == 2) issue_fn(2);
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) pacc[rf][nf] = mma(awh[nf], axl[rf], pacc[rf][nf]);
            if (nf == 0) issue_fn(3);
        }
        if (FFS_CPRIO) { pin(); __builtin_amdgcn_s_setprio(0); }
    };
    auto b_load = [&](int sb, int slot_i, auto issue_fn_) {
        auto issue_fn = [&](int u) {
            if (FFS_ONEHALF) { if (u < 3) { issue_fn_(2 * u); issue_fn_(2 * u + 1); } else issue_fn_(6); return; }  // 0 1 | 2 3 | 4 5 | 6 (a seventh piece exists when step s + 3 is an A-step)
            if (FFS_DMA_IN_C == 0 || (FFS_DMA_IN_C == 2 && u < 2)) issue_fn_(u);
        };
        const int so = slot_off(slot_i);
        issue_fn(0);
        pin();
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) bwh[nf] = b_w(so, nf, 0);
        pin();
        issue_fn(1);
        pin();
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) bwl[nf] = b_w(so, nf, 1);
        pin();
        issue_fn(2);
        pin();
        if ((sb & 1) == 0) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) { bgh[rf] = b_g(sb >> 1, rf, 0); bgl[rf] = b_g(sb >> 1, rf, 1); }
        }
        issue_fn(3);
    };
    auto b_compute = [&](int sb, auto issue_fn_, auto side) {  // side(k): VALU work behind the k-th MFMA (k = 0 .. 26)
        auto issue_fn = [&](int u) { if (!FFS_ONEHALF && (FFS_DMA_IN_C == 1 || (FFS_DMA_IN_C == 2 && u >= 2))) { pin(); issue_fn_(u); pin(); } };
        const int half = sb & 1;
        int k = 0;
        if (FFS_CPRIO) __builtin_amdgcn_s_setprio(FFS_CPRIO);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) { acc[rf][half * 3 + nf] = mma(bwh[nf], bgh[rf], acc[rf][half * 3 + nf]); side(k++); }
            if (rf == 1) issue_fn(0);
        }
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) { acc[rf][half * 3 + nf] = mma(bwl[nf], bgh[rf], acc[rf][half * 3 + nf]); side(k++); }
            if (rf == 1) issue_fn(1);
        }
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) { acc[rf][half * 3 + nf] = mma(bwh[nf], bgl[rf], acc[rf][half * 3 + nf]); side(k++); }
            if (nf == 0) { issue_fn(2); issue_fn(3); }
        }
        if (FFS_CPRIO) { pin(); __builtin_amdgcn_s_setprio(0); }
    };

    // rows past M read row M - 1; nothing of them is ever stored (computed where it is used: as an array it ended up in scratch)
    auto mrow = [&](int rf) { const int m = m0 + rows0 + rf * 16; return m < p.M ? m : p.M - 1; };
    // LayerNorm of the 96 x 384 block in the accumulators (row statistics in registers, one LDS exchange between the column
    // quarters through the G region, which must be out of use): h_dst <- LN(acc) gamma + beta in the split format; x_dst (or
    // NULL) <- acc
    auto layernorm_rows = [&](const float* gamma, const float* beta, float* x_dst, void* h_dst) {
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
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n), b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                const bool live = m0 + rows0 + rf * 16 < p.M;
                const size_t off = (size_t)(m0 + rows0 + rf * 16) * E + n;
                const f32x4 v = acc[rf][cf];
                float mu = mean[rf];
                const float rs = rstd[rf];
                asm("" : "+v"(mu));  // (a second, opaque copy: with the same value as in the variance pass the compiler keeps all 72 differences v - mean alive from there to here)
                f32x4 hv = {(v[0] - mu) * rs * g[0] + b[0], (v[1] - mu) * rs * g[1] + b[1], (v[2] - mu) * rs * g[2] + b[2],
                            (v[3] - mu) * rs * g[3] + b[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { float t = hv[j]; split_pin(t); hv[j] = t; }  // (pp_split.h)
                if (x_dst && live) *reinterpret_cast<f32x4*>(x_dst + off) = v;
#ifndef FFS_PAIR_STORE
#define FFS_PAIR_STORE 1  // dev A/B switch: 0 two 8-byte stores per lane and fragment
#endif
#ifndef FFS_HS_NT
#define FFS_HS_NT 0  // dev: ln2 rows (x_dst == nullptr) stored non-temporally
#endif
                if (FFS_HS_NT && !x_dst) split_store4_rowpair_nt(h_dst, off, hv, live);
                else if (FFS_PAIR_STORE) split_store4_rowpair(h_dst, off, hv, live);  // (lanes f_kg, f_kg ^ 1 hold the halves of a 16-byte chunk)
                else if (live) split_store4(h_dst, off, hv);
            }
        }
    };

    if constexpr (PROJ) {
        stamp();
        stamp();
        // ================= attention output projection + residual, then ln2:  acc <- x + att Wp^T + bp ;  h <- LN2(acc).
        // 24 steps shaped like the B-steps (wave tile 48 rows x 48 outputs of one column half, 27 MFMAs): step s = 2 kb + half
        // takes the Wp half block (192 lines, pre-packed like the W2 blocks) from ring slot s & 3 and the k-block kb of the
        // attention rows from G buffer kb & 3 (96 lines, fetched with the even steps by waves 0-3). Two steps in flight, one
        // barrier per step, all waves in the same phase: the phase is ~6 % of the launch, the FFN machinery below is not spent
        // on it. The ln2 rows then go out to `h` (global, L2) and come back as the streamed row operand of the A-steps.
        // The residual rows (147 KB per workgroup) are requested FIRST, ahead of every DMA piece, straight into the accumulators
        // (the projection adds onto them). They must not trickle in between the pieces: a plain load that is YOUNGER than an
        // LDS-DMA piece can retire before that piece's LDS write does (the vmcnt decrements of the two kinds are not ordered
        // with respect to each other), so a counted wait that allows "the loads issued since" to be outstanding lets a piece
        // through that has not landed - measured: 5 - 26 of 80 launches wrong when a second stream shares the chip
        // (scripts/micro/ffs_stress_two_streams.py), none alone. Older plain loads are safe: the first counted wait covers them.
        // (Held in a second register set until the end of the phase they cost 72 registers: 34 spilled.)
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int cf = 0; cf < 6; ++cf) {
                const int n = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4;
                acc[rf][cf] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mrow(rf) * E + n);
            }
        auto issue_p = [&](int s) {
            if (DBG & 8) return;
            const bool live = s < 2 * KB;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wproj), 0, live ? p.wproj_bytes : 0u, 0x00020000);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ring + (s & 3) * SLOTB + (3 * wv + u) * 1024), 16, v_w,
                                                         (live ? s : 0) * B_BLOCK + (3 * wv + u) * 1024, 0, 0);
            if ((s & 1) == 0 && rg == 0) {
                const int kb = s >> 1;
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.att), 0, live ? p.att_bytes : 0u, 0x00020000);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(smem + OFF_G + (kb & 3) * G_KB + (3 * wv + u) * 1024), 16, v_x,
                                                             (live ? kb : 0) * 128 + u * 8 * E * 4, 0, 0);
            }
        };
#ifndef FFS_PPRIO
#define FFS_PPRIO 1  // 1 = waves 4-7 (the younger half, which loses the matrix-pipe arbitration) at priority 1 during the projection steps: 201.9 -> 200.2 us; 2 = for the whole kernel (no further gain); 0 = off
#endif
        if (FFS_PPRIO && rg == 1) __builtin_amdgcn_s_setprio(1);
        issue_p(0);
        issue_p(1);
#pragma unroll 1
        for (int kp = 0; kp < 2 * KB / 4; ++kp) {  // (rolled: fully unrolled the 24 steps cost 34 spilled registers)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * kp + q;
                // step s has landed when only this wave's pieces of step s + 1 are outstanding
                __builtin_amdgcn_sched_barrier(0);
                stamp();
                if (rg == 0) { if (q & 1) FFS_WAIT(6); else FFS_WAIT(3); } else FFS_WAIT(3);
                __builtin_amdgcn_s_barrier();
                stamp();
                __builtin_amdgcn_sched_barrier(0);
                issue_p(s + 2);
                const int so = slot_off(q);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) { bwh[nf] = b_w(so, nf, 0); bwl[nf] = b_w(so, nf, 1); }
                if ((q & 1) == 0) {
                    const int j = (s >> 1) & 3;
#pragma unroll
                    for (int rf = 0; rf < 3; ++rf) { bgh[rf] = b_g(j, rf, 0); bgl[rf] = b_g(j, rf, 1); }
                }
                b_compute(q & 1, [](int) {}, [](int) {});
            }
        }
        // + bp (after the sums, as residual + (sum + bias) rounds closest to the reference's x + proj(...))
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bp + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][cf] += bv;
        }
        if (FFS_PPRIO == 1 && rg == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | (0));  // vmcnt(0) lgkmcnt(0): the zero fillers of steps 24, 25 have landed too
        __syncthreads();
        stamp();
        layernorm_rows(p.gamma2, p.beta2, nullptr, const_cast<void*>(p.h));
        stamp();
        // the rows must be in L2 before any wave's DMA asks for them (a store counts in vmcnt until the L2 has acknowledged it)
        __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | (0));
#ifndef FFS_HS_FENCE
#define FFS_HS_FENCE 0
#endif
        if (FFS_HS_FENCE & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (FFS_HS_FENCE & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        stamp();
        stamp();
    }

    // ---- prologue: b1 of the first chunk and (without the projection phase) the residual rows, all OLDER than every DMA piece:
    // landed at the first counted wait; then the DMA of steps 0 - 2
    load_b1(0);
    if constexpr (!PROJ) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int cf = 0; cf < 6; ++cf) {
                const int n = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4;
                acc[rf][cf] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mrow(rf) * E + n);
            }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) issue_step(0, t - NA);
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = b1v[nf];
    // step 0 has landed when at most the pieces of steps 1, 2 are outstanding
    wait_and_barrier<2 * n_ops(0, 0), 2 * n_ops(0, 1)>(rg);
    if (rg == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one segment behind

    // ---- peeled A-steps of the first chunk
#pragma unroll
    for (int kt = 0; kt < NA; ++kt) {
        a_load(kt & 3, [&](int u) { issue_piece(0, kt - NA + 3, u); });
        sync_l([&]() { if (FFS_ONEHALF) FFS_WAIT(63); else if (FFS_DMA_IN_C == 1) FFS_WAIT(3); else if (FFS_DMA_IN_C == 2) FFS_WAIT(3 + 2); else FFS_WAIT(2 * 3); });
        a_compute([&](int u) { issue_piece(0, kt - NA + 3, u); });
        sync_c([&]() { if (FFS_ONEHALF) FFS_WAIT(14); else FFS_WAIT(2 * 4); });
    }
    // + b2; the first chunk's GELU has no other wave half's MFMAs... it runs beside the OTHER half's segments all the same,
    // but this half's next L segment waits for it (once per launch)
#pragma unroll
    for (int cf = 0; cf < 6; ++cf) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) acc[rf][cf] += bv;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) gelu_pairs(2 * c, 2, false);
    write_g(0, 6);
    load_b1(1);
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8) | (0));  // vmcnt(0): b2, b1 in; (DMA pieces too - once per launch)

    for (int it = 0; it < nchunks; ++it) {
        if (it > 0) {
            pold[0] = pacc[2][0];
            pold[1] = pacc[2][1];
        }
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = b1v[nf];
        // ================= A-steps of chunk it + 1. Their L segments carry the GELU leftovers of chunk it (B-steps below):
        // fragments 0-3 are stored at step 0, pairs 8..11 (fragments 4, 5, kept in pold) run at steps 1, 3, 5, 7, stored at step 8
#pragma unroll
        for (int kt = 0; kt < NA; ++kt) {
            a_load(kt & 3, [&](int u) { issue_piece(it, kt + 3, u); });
            if (it > 0) {
                if (kt == 0) write_g(0, 4);
                if (kt >= 1 && kt <= 7 && (kt & 1)) gelu_pairs(8 + (kt >> 1), 1, true);
                if (kt == 8) write_g(4, 6);
            }
            // pieces of steps kt + 2, kt + 3: A A up to 8, A B at 9, B B at 10, 11 (the first iteration's vmcnt(0) above makes
            // every count an upper bound there)
            sync_l([&]() { if (FFS_ONEHALF) FFS_WAIT(63); else if (FFS_DMA_IN_C == 1) FFS_WAIT(3); else if (FFS_DMA_IN_C == 2) FFS_WAIT(3 + 2); else if (kt <= 8) FFS_WAIT(2 * 3); else if (kt == 9) FFS_WAIT(3 + 3); else FFS_WAIT(6); });
            a_compute([&](int u) { issue_piece(it, kt + 3, u); });
            sync_c([&]() { if (FFS_ONEHALF) { if (kt <= 8) FFS_WAIT(14); else if (kt == 9) FFS_WAIT(13); else FFS_WAIT(12); } else if (kt <= 8) FFS_WAIT(2 * 4); else if (kt == 9) FFS_WAIT(4 + 3); else FFS_WAIT(6); });
        }
        // ================= B-steps of chunk it; their L segments carry one GELU pair of chunk it + 1 each (pairs 0..7)
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {
            const int t = NA + sb;
            if (sb == 0) load_b1(it + 2);  // (in front of this segment's pieces: two more operations outstanding at steps 12, 13)
            b_load(sb, t & 3, [&](int u) { issue_piece(it, t + 3, u); });
            if (!FFS_GELU_IN_C) gelu_pairs(sb, 1, false);
            // pieces of steps t + 2, t + 3: B B up to t = 16, B A' at 17, A' A' at 18, 19
            // (the two b1 loads of step 12 get NO slack in the counts: as younger plain loads they may retire before the pieces
            // these waits are for - see the projection phase)
            const int ex = 0;
            sync_l([&]() { if (FFS_ONEHALF) FFS_WAIT(63); else if (FFS_DMA_IN_C == 1) FFS_WAIT(3); else if (FFS_DMA_IN_C == 2) FFS_WAIT(3 + 2); else if (t <= 16) { if (ex) FFS_WAIT(6 + 2); else FFS_WAIT(6); } else if (t == 17) FFS_WAIT(3 + 3); else FFS_WAIT(2 * 3); });
            b_compute(sb, [&](int u) { issue_piece(it, t + 3, u); }, [&](int k) { if (FFS_GELU_IN_C && k < 24) { pin(); gelu_stage(k, sb); pin(); } });
            sync_c([&]() { if (FFS_ONEHALF) { if (t <= 16) FFS_WAIT(12); else if (t == 17) FFS_WAIT(13); else FFS_WAIT(14); } else if (t <= 16) { if (ex) FFS_WAIT(6 + 2); else FFS_WAIT(6); } else if (t == 17) FFS_WAIT(3 + 4); else FFS_WAIT(2 * 4); });
        }
    }
    if (rg == 0) __builtin_amdgcn_s_barrier();  // waves 0-3 are one segment ahead
    // ---- LayerNorm epilogue: G is out of use, its region carries the statistics exchange; the ring may still receive the
    // out-of-bounds fillers issued past the last chunk
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | (0));  // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    layernorm_rows(p.gamma, p.beta, p.x_out, p.h_out);
}

__global__ __launch_bounds__(THREADS, 2) void ffn_split_kernel(const Params p) { ffn_split_body<false>(p); }
__global__ __launch_bounds__(THREADS, 2) void proj_ffn_split_kernel(const Params p) { ffn_split_body<true>(p); }

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

// Packs Wp (384, 384), split row-major, into the stream of the projection phase: 24 blocks [192 outputs][128 B], block
// s = 2 kb + half = rows 192 half .. of k-block kb, lines swizzled like the W2 blocks.
__global__ void pack_proj_kernel(const char* __restrict__ wp, char* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 16-byte chunk of the output
    if (idx >= 2 * KB * (B_BLOCK / 16)) return;
    const int s = idx / (B_BLOCK / 16), q = idx % (B_BLOCK / 16), line = q >> 3, pc = q & 7;
    const int n = (s & 1) * (E / 2) + line, kb = s >> 1;
    *reinterpret_cast<u32x4*>(out + (size_t)idx * 16) =
        *reinterpret_cast<const u32x4*>(wp + ((size_t)n * E + kb * 32) * 4 + ((pc ^ (line & 7)) << 4));
}

unsigned long long* g_trace = nullptr;
int launch_dma_form(const Params& p, bool proj, hipStream_t s);  // pp_ffn_dma.hip

template <bool PROJ>
static int launch(const Params& p, hipStream_t s) {
    if (option("ffn_dma_waves") != 0) return launch_dma_form(p, PROJ, s);
    auto kern = PROJ ? proj_ffn_split_kernel : ffn_split_kernel;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(kern, dim3((p.M + BM - 1) / BM), dim3(THREADS), LDS, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
}  // namespace ffs
}  // namespace pp

#if FFS_DBG & 512
extern "C" void pp_ffs_set_trace(void* buf) { pp::ffs::g_trace = reinterpret_cast<unsigned long long*>(buf); }
#endif

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
    p.trace = ffs::g_trace;
    return ffs::launch<false>(p, reinterpret_cast<hipStream_t>(stream));
}

extern "C" long long pp_proj_split_packed_bytes(int E) {
    using namespace pp;
    return E == ffs::E ? (long long)2 * ffs::KB * ffs::B_BLOCK : -1;
}

extern "C" int pp_proj_split_pack_weights(const void* wp, void* packed, int E, void* stream) {
    using namespace pp;
    PP_REQUIRE(wp && packed, PP_ERR_INVALID_ARG, "pp_proj_split_pack_weights: NULL argument");
    PP_REQUIRE(E == ffs::E, PP_ERR_UNSUPPORTED, "pp_proj_split_pack_weights: built for embed dim 384 (ViT-S)");
    const int total = 2 * ffs::KB * (ffs::B_BLOCK / 16);
    hipLaunchKernelGGL(ffs::pack_proj_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const char*>(wp), reinterpret_cast<char*>(packed));
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_proj_ffn_split_residual_layernorm(const void* att, const void* wproj_packed, const float* bproj,
                                                    const float* gamma2, const float* beta2, void* h_scratch,
                                                    const void* w_packed, const float* b1, const float* b2,
                                                    const float* residual, float* x_out, const float* gamma, const float* beta,
                                                    float eps, void* h_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(att && wproj_packed && bproj && gamma2 && beta2 && h_scratch && w_packed && b1 && b2 && residual && x_out && gamma &&
                   beta && h_out,
               PP_ERR_INVALID_ARG, "pp_proj_ffn_split_residual_layernorm: NULL argument");
    PP_REQUIRE(E == ffs::E, PP_ERR_UNSUPPORTED, "pp_proj_ffn_split_residual_layernorm: built for embed dim 384 (ViT-S)");
    PP_REQUIRE(M > 0 && F > 0 && F % ffs::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_proj_ffn_split_residual_layernorm: hidden width must be a positive multiple of 128");
    PP_REQUIRE((size_t)M * E * 4 < ffs::OOB && (size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES < ffs::OOB, PP_ERR_UNSUPPORTED,
               "pp_proj_ffn_split_residual_layernorm: operand exceeds 2 GiB");
    PP_REQUIRE(h_scratch != att && h_scratch != h_out, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_residual_layernorm: h_scratch must not alias the attention rows or h_out");
    ffs::Params p{};
    p.h = h_scratch;
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
    p.trace = ffs::g_trace;
    p.att = att;
    p.wproj = wproj_packed;
    p.bp = bproj;
    p.gamma2 = gamma2;
    p.beta2 = beta2;
    p.att_bytes = p.h_bytes;
    p.wproj_bytes = (unsigned)(2 * ffs::KB * ffs::B_BLOCK);
    return ffs::launch<true>(p, reinterpret_cast<hipStream_t>(stream));
}

// The projection + FFN launch inside a chain of layers whose LayerNorm in front of the qkv projection is folded into that projection
// (pp_qkv_attention_split_folded): the residual rows arrive and / or leave in the operand format, and with fold_out the final LayerNorm of this
// launch is NOT applied - the rows leave once (h_out: x in the operand format) with (mean, rstd) per row (stats_out). Twelve-wave paired kernel only.
extern "C" int pp_proj_ffn_split_folded(const void* att, const void* wproj_packed, const float* bproj, const float* gamma2, const float* beta2,
                                        void* h_scratch, const void* w_packed, const float* b1, const float* b2, const void* residual,
                                        int residual_format, int fold_out, float* x_out, const float* gamma, const float* beta, float eps, void* h_out,
                                        float* stats_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(att && wproj_packed && bproj && gamma2 && beta2 && h_scratch && w_packed && b1 && b2 && residual && h_out, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: NULL argument");
    PP_REQUIRE(residual_format == PP_OUT_F32 || residual_format == PP_OUT_SPLIT, PP_ERR_INVALID_ARG, "pp_proj_ffn_split_folded: residual_format is PP_OUT_F32 or PP_OUT_SPLIT");
    PP_REQUIRE(fold_out ? stats_out != nullptr : (x_out && gamma && beta), PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: fold_out needs stats_out; without it x_out, gamma and beta are required");
    PP_REQUIRE(E == ffs::E, PP_ERR_UNSUPPORTED, "pp_proj_ffn_split_folded: built for embed dim 384 (ViT-S)");
    PP_REQUIRE(M > 0 && F > 0 && F % ffs::CHUNK == 0 && (F / ffs::CHUNK) % 2 == 0, PP_ERR_UNSUPPORTED,
               "pp_proj_ffn_split_folded: hidden width must be an even number of 128-column chunks (the paired twelve-wave kernel)");
    PP_REQUIRE(option("ffn_dma_waves") != 0, PP_ERR_UNSUPPORTED, "pp_proj_ffn_split_folded: needs the twelve-wave kernel (option ffn_dma_waves)");
    PP_REQUIRE((size_t)M * E * 4 < ffs::OOB && (size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES < ffs::OOB, PP_ERR_UNSUPPORTED,
               "pp_proj_ffn_split_folded: operand exceeds 2 GiB");
    PP_REQUIRE(h_scratch != att && h_scratch != h_out && h_scratch != residual, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: h_scratch must not alias the attention rows, the residual rows or h_out");
    ffs::Params p{};
    p.h = h_scratch;
    p.wpack = w_packed;
    p.b1 = b1;
    p.b2 = b2;
    p.residual = reinterpret_cast<const float*>(residual);
    p.x_out = x_out;
    p.gamma = gamma;
    p.beta = beta;
    p.h_out = h_out;
    p.M = M;
    p.F = F;
    p.h_bytes = (unsigned)((size_t)M * E * 4);
    p.w_bytes = (unsigned)((size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES);
    p.eps = eps;
    p.trace = ffs::g_trace;
    p.att = att;
    p.wproj = wproj_packed;
    p.bp = bproj;
    p.gamma2 = gamma2;
    p.beta2 = beta2;
    p.att_bytes = p.h_bytes;
    p.wproj_bytes = (unsigned)(2 * ffs::KB * ffs::B_BLOCK);
    p.res_split = residual_format == PP_OUT_SPLIT;
    p.fold_out = fold_out != 0;
    p.stats_out = stats_out;
    return ffs::launch_dma_form(p, true, reinterpret_cast<hipStream_t>(stream));
}

