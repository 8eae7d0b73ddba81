// Fused ViT feed-forward block for gfx950 (bf16 operands):
//     x <- x + GELU(h W1^T + b1) W2^T + b2 ;   h_next <- LayerNorm(x)
// (mmpretrain TransformerEncoderLayer [3P]: x = ffn(ln2(x), identity = x), FFN = Linear - GELU(erf) - Linear, followed by
// the next layer's ln1 or the final ln1). Done as two GEMM launches, the 4x-wide hidden activation (75 MB at bs 64)
// is written to HBM and read back; it is a quarter of all the bytes a ViT-S layer moves. Here it never leaves the CU.
//
//   * one workgroup owns 96 complete token rows (grid = M / 96 = one workgroup per CU at bs 64 with flip test),
//     512 threads = 8 waves = two per SIMD, wave (rg, cg): rows 48 rg .. +47, column quarter cg;
//   * the LayerNorm-ed input rows h sit in LDS for the whole kernel (72 KiB, MFMA operand image);
//   * the hidden layer is processed in chunks of 128 units, ten steps per chunk:
//       phase A  P = h W1[chunk]^T        6 steps of k = 64, wave tile 48 rows x 32 units
//                G = GELU(P + b1) -> bf16 -> LDS (24 KiB) as the operand of phase B
//       phase B  acc += G W2[:, chunk]^T   4 steps of k = 32, wave tile 48 rows x 96 outputs
//     software-pipelined across chunks: the loop body is [phase A of chunk c+1 | phase B of chunk c], and the GELU
//     of chunk c is spread over the six phase-A steps of chunk c+1, one accumulator fragment per step, so its VALU
//     instructions sit in the shadow of that chunk's MFMAs (sched_group_barrier pins the interleave);
//   * W1 / W2 tiles arrive by LDS-DMA through ONE ring of eight 8 KiB slots (a phase-A tile is two slots, a phase-B
//     tile three; 24 slots per chunk, so ring positions are compile-time constants). Every wave issues one DMA
//     instruction per slot and waits with a COUNTED vmcnt, so five slots (40 KiB) stay in flight at all times;
//   * operand fragments are double-buffered in registers: step s issues the ds_reads of tile s+1, then runs the
//     MFMAs of tile s on fragments that were read a whole step earlier - no MFMA waits on its own LDS read, and a
//     tile's ring slots are free for the DMA as soon as the step that consumes it begins (one barrier per step);
//   * the 96 x 384 output accumulators start from residual + b2 and end in the LayerNorm epilogue (row statistics in
//     registers, one LDS exchange between the column quarters).
// LDS: 72 KiB h + 24 KiB G + 64 KiB ring = 160 KiB, the whole CU.
// GELU uses a clamped odd polynomial for erf (|error| < 1.8e-4, a factor 60 under bf16 resolution): libdevice erff costs
// as many VALU cycles as the MFMAs of the whole block.
#include "pp_common.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

namespace mlp {

#ifndef MLP_DBG
#define MLP_DBG 0
#endif
// dev ablation switches (scripts/micro/mlp_ablate.sh), 0 in the product build: 2 no GELU, 4 no MFMA, 8 no DMA,
// 16 no LDS fragment reads, 32 no barriers, 64 no b1 loads, 512 per-step time stamps, 1024 no qkv stores, 2048 no qkv staging / stores at all
constexpr int DBG = MLP_DBG;
#ifndef ATT_DBG
#define ATT_DBG 0  // dev ablations of the attention rounds: 1 no v_exp, 2 no V reads, 4 no K reads, 8 no P V MFMAs (wrong results)
#endif
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_VALU = 0x002, SG_MFMA = 0x008, SG_VMEM = 0x010, SG_DS_READ = 0x100;
constexpr int BM = 96, E = 384, CHUNK = 128;
constexpr int WAVES = 8, THREADS = 64 * WAVES;
constexpr int ROW_BYTES = 128;
constexpr int HS_KB = BM * ROW_BYTES;             // 12 KiB: 96 rows x 64 k
constexpr int OFF_HS = 0;                         // [6][96][128 B]
constexpr int OFF_GS = 6 * HS_KB;                 // [2][96][128 B]
constexpr int OFF_RING = OFF_GS + 2 * HS_KB;
constexpr int SLOT = 8192, NSLOT = 8;
constexpr int LDS = OFF_RING + NSLOT * SLOT;      // 163 840 B
constexpr int KT1 = 6, KT2 = 4;                   // steps of phase A (k 64) and phase B (k 32)
constexpr int SLOTS_PER_CHUNK = 2 * KT1 + 3 * KT2;  // 24
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS == 160 * 1024, "LDS map");
static_assert(SLOTS_PER_CHUNK % NSLOT == 0, "ring positions must repeat per chunk");

extern unsigned long long* g_trace;
struct Params {
    const __bf16* h;       // [M, 384] LayerNorm-ed block input (PROJ: the attention output rows)
    const __bf16* Wp;      // PROJ: [384, 384] attention output projection
    const float* bp;       // PROJ: [384]
    const float* gamma2;   // PROJ: LayerNorm in front of the FFN (ln2)
    const float* beta2;
    const __bf16* qkv_in;  // ATT: [M, 1152] q | k | v of THIS layer (row-major, as the qkv Linear / the previous launch wrote it)
    float scale_log2e;     // ATT: head_dim^-0.5 * log2(e)
    const __bf16* Wq;      // QKV: [1152, 384] qkv projection of the NEXT layer
    const float* bq;       // QKV: [1152]
    __bf16* qkv;           // QKV: [M, 1152] output
    const __bf16* W1;      // [F, 384]
    const float* b1;       // [F]
    const __bf16* W2;      // [384, F]
    const float* b2;       // [384]
    const float* residual; // fp32 [M, 384] (may alias x_out)
    float* x_out;          // fp32 [M, 384]
    const float* gamma;
    const float* beta;
    __bf16* h_out;         // [M, 384] LayerNorm(x_out)
    int M, F;
    unsigned h_bytes, w1_bytes, w2_bytes, wp_bytes, wq_bytes;
    float eps;
    unsigned long long* trace;  // dev only (MLP_DBG & 512): per-step time stamps of block 0, waves 0 and 4
};

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) = x (0.5 + t W(t^2)), t = clamp(x, -4.2, 4.2), where t W(t^2) is a degree-15 odd
// minimax polynomial for 0.5 erf(t / sqrt 2), constrained to reach exactly 0.5 at |t| = 4.2 so that the function
// saturates to x and to 0 outside. Absolute error < 1.8e-4 for every x, relative error < 6.2e-5 for x > 0.05 - a factor
// 60 under bf16 resolution, which is what the result is rounded to. Twelve plain fp32 instructions and no
// transcendental. Plain (unpacked) on purpose - this file is built with -fno-slp-vectorize: the GELU of the previous
// chunk is issued between the MFMAs of the current one, and beside MFMAs a v_pk_fma_f32 costs about 22 cycles more
// than the two v_fma_f32 it replaces.
__device__ __forceinline__ float gelu_fast(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
    const float s = t * t;
    float q = __builtin_fmaf(s, -1.141911177e-09f, 9.614189360e-08f);
    q = __builtin_fmaf(s, q, -3.508876526e-06f);
    q = __builtin_fmaf(s, q, 7.374335597e-05f);
    q = __builtin_fmaf(s, q, -1.005266667e-03f);
    q = __builtin_fmaf(s, q, 9.529921441e-03f);
    q = __builtin_fmaf(s, q, -6.599143966e-02f);
    q = __builtin_fmaf(s, q, 3.987765802e-01f);
    return x * __builtin_fmaf(t, q, 0.5f);
}
__device__ __forceinline__ bf16x4 gelu4_bf16(f32x4 v) {
    return bf16x4{(__bf16)gelu_fast(v[0]), (__bf16)gelu_fast(v[1]), (__bf16)gelu_fast(v[2]), (__bf16)gelu_fast(v[3])};
}

// slots consumed by step s of a chunk (0..5 phase A, 6..9 phase B) and the first slot of step s
__host__ __device__ constexpr int step_slots(int s) { return (s % 10) < KT1 ? 2 : 3; }
__host__ __device__ constexpr int step_first(int s) { return s < KT1 ? 2 * s : 2 * KT1 + 3 * (s - KT1); }

// wait until at most N of this wave's vector-memory operations are outstanding and every LDS read has returned,
// then meet the other waves
template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_sched_barrier(0);  // nothing of the previous step may sink below, nothing of the next may rise above
    if (!(DBG & 32)) {
        // s_waitcnt vmcnt(N) lgkmcnt(0) as a builtin, not inline asm: the compiler's own wait-count bookkeeping sees it
        // and does not re-wait for the fragment reads in front of the MFMAs that use them (gfx9 encoding:
        // vmcnt [3:0] + [15:14], expcnt [6:4] = 7 (none), lgkmcnt [11:8])
        __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
}

// qkv tile stream of the tail: step t = 6 b + kt (column block b of 192 outputs, k-step kt of 64), three slots of 64
// weight rows each: slot q = 3 t + third holds rows 192 b + 64 third + (0..63), k in [64 kt, +64), plain 128-byte lines
// like a W1 slot. Past the last slot the DMA is issued out of bounds (a plain function: a value-returning lambda for this
// inside the kernel template made the host pass drop the kernel stubs without a diagnostic).
__device__ __forceinline__ int rot_mod(int v, int rot, int n) {  // (v + rot) mod n for v, rot in [0, n)
    const int r = v + rot;
    return r >= n ? r - n : r;
}
// The column blocks are visited in a per-workgroup rotation (block (t / 6 + rot) % 6 at step t), like the hidden chunks of
// the FFN: every CU streams the same Wq, in lockstep they would all pull the same lines out of their L2 at once.
__device__ __forceinline__ unsigned wq_slot_offset(int q, unsigned lane_off, int rot) {
    const int t = q / 3;
    const int blk = rot_mod(t / 6, rot, 6);
    return q < 108 ? (unsigned)(blk * 192 + (q % 3) * 64) * (unsigned)(E * 2) + (unsigned)((t % 6) * 128) + lane_off : OOB;
}

// PROJ = true puts the attention output projection in front:  x' = x + a Wp^T + bp ;  h = LayerNorm2(x')  and then the
// block above on (x', h) - the 96 x 384 accumulators that end the projection ARE the residual the FFN accumulates on,
// so x' and h never travel to HBM. The projection is 12 more steps of the phase-B kind (Wp tiles of 384 outputs x
// 32 k = 3 ring slots) at the head of the same slot stream.
// QKV = true appends the next layer's qkv projection:  qkv = h_out Wq^T + bq  for the 96 rows, six column blocks of
// 192 in two accumulator sets (the block's accumulators reused), 36 steps of k = 64 at the tail of the slot stream; h_out
// then goes to LDS instead of HBM (p.h_out may be NULL).
// ATT = true puts the attention itself in front of the projection: the 96 rows are half of a 192-token sequence; for
// each of the 12 heads K and V of the whole sequence come in by LDS-DMA (three heads in rotation: two in the ring's
// region, which the weight stream only needs afterwards, one in the G region), S^T = K Q^T, softmax in registers,
// O^T = V^T P^T with V read through the transposing ds_read_b64_tr_b16, and the normalised output rows go straight into
// the LDS image the projection reads - the attention output never exists in HBM and a ViT layer is ONE launch. The
// residual rows load under this phase, the first weight slots under its last head.
template <bool PROJ, bool QKV, bool ATT>
__global__ __launch_bounds__(THREADS, 2) void mlp_res_ln_kernel(const Params p) {
    static_assert(!ATT || PROJ, "the attention phase feeds the projection");
    constexpr int PRE = PROJ ? 36 : 0;  // ring slots streamed before the first phase A
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    // Row tile of this workgroup. Hardware hands consecutive block ids to the 8 XCDs in turn; with ATT the two 96-row halves
    // of a 192-token sequence read the same K and V, so they are placed on ONE XCD (shared L2): XCD x takes the contiguous
    // run of tiles [x * G / 8, (x + 1) * G / 8). Without this both halves fetched K / V from HBM on their own
    // (r01: 262 MB per launch of counter traffic against 186 MB algorithmic).
    int tile = blockIdx.x;
    if ((gridDim.x & 15) == 0) tile = (tile & 7) * (gridDim.x >> 3) + (tile >> 3);
    const int m0 = tile * BM;

    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.h), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.W1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.W2), 0, p.w2_bytes, 0x00020000);

    // ---- DMA addressing. One instruction moves 8 lines of 128 B; lane i lands at byte 16 i, so the XOR swizzle is
    // applied to the SOURCE: lane (line l, physical chunk pc) fetches logical chunk pc ^ (l & 7).
    const int d_line = wv * 8 + (lane >> 3);            // line inside a slot (0..63)
    const int d_lc = (lane & 7) ^ (lane >> 3);          // logical 16-byte chunk (d_line & 7 == lane >> 3)
    // W1 slot (kt, half): line l holds unit 64 half + l, k in [64 kt, +64)
    const unsigned w1_lane = (unsigned)d_line * (E * 2) + (unsigned)(d_lc << 4);
    // W2 slot (j, third): line l' = 64 third + l holds outputs l' (chunks 0..3) and l' + 192 (chunks 4..7), k in [32 j, +32)
    const unsigned w2_row_bytes = (unsigned)p.F * 2u;
    const unsigned w2_lane = (unsigned)(d_line + 192 * (d_lc >> 2)) * w2_row_bytes + (unsigned)((d_lc & 3) << 4);
    // Wp slot (j, third): the same two-rows-per-line image, row stride 384 elements
    const unsigned wp_lane = (unsigned)(d_line + 192 * (d_lc >> 2)) * (E * 2) + (unsigned)((d_lc & 3) << 4);
    const __amdgpu_buffer_rsrc_t wp_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(PROJ ? p.Wp : p.W1), 0, PROJ ? p.wp_bytes : p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wq_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(QKV ? p.Wq : p.W1), 0, QKV ? p.wq_bytes : p.w1_bytes, 0x00020000);

    const int nchunks = p.F / CHUNK;
    // Every workgroup walks the hidden chunks in a different rotation: all 256 CUs stream the SAME weights, and in
    // lockstep they would all hit the same L2 lines at the same moment.
    // (the rotation index is the workgroup's rank inside its XCD - blockIdx.x >> 3 - so that the CUs behind one L2 are
    // spread over all rotations; blockIdx.x % nchunks gave the 32 CUs of an XCD only three distinct ones)
    const int xcd_rank = (int)(blockIdx.x >> 3);
    const int c_rot = xcd_rank % nchunks;
    const int p_rot = xcd_rank % 12;  // k-steps of the projection (a sum: any order)
    const int q_rot = xcd_rank % 6;   // column blocks of the qkv tail
    // (tried and dropped: a second level - the k-steps INSIDE a chunk's phase A / phase B rotated too, for the CUs of an
    // XCD that share a chunk rotation - 169 us against 143: the run-time address arithmetic lands in the FFN loop)
    auto chunk_of = [&](int i) { const int c = i + c_rot; return c >= nchunks ? c - nchunks : c; };

    char* const ring = smem + OFF_RING;
    // One slot of the phase-A tile sequence of the chunk visited ci-th (q = 2 kt + half) / of its phase-B tile
    // sequence (q = 3 j + third) into ring position pos. Past the last chunk the DMA is issued out of bounds (it
    // writes zeros) so that every wave's vmcnt arithmetic stays the same.
    auto issue_w1 = [&](int ci, int q, int pos) {
        if (DBG & 8) return;
        const bool live = ci < nchunks;
        const unsigned vo = (unsigned)(chunk_of(live ? ci : 0) * CHUNK + (q & 1) * 64) * (E * 2) + (unsigned)((q >> 1) * 128) + w1_lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, (lds_ptr_t)(ring + pos * SLOT + wv * 1024), 16, live ? vo : OOB, 0, 0, 0);
    };
    auto issue_w2 = [&](int ci, int q, int pos) {
        if (DBG & 8) return;
        const bool live = ci < nchunks;
        const unsigned vo = (unsigned)((q % 3) * 64) * w2_row_bytes + (unsigned)((chunk_of(live ? ci : 0) * CHUNK + 32 * (q / 3)) * 2) + w2_lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (lds_ptr_t)(ring + pos * SLOT + wv * 1024), 16, live ? vo : OOB, 0, 0, 0);
    };
    // The slot stream, in consumption order: A(0) | A(1) B(0) | A(2) B(1) | ... - twelve slots for the peeled first
    // phase A, then 24 per loop iteration it (phase A of chunk it + 1, phase B of chunk it). g is the position in the
    // stream relative to the start of iteration `it` (negative = the peeled phase, below -12 = the projection);
    // ring position = stream index & 7.
    auto issue_wp = [&](int q, int pos) {
        if (DBG & 8) return;
        const int j = rot_mod(q / 3, p_rot, 12);
        const unsigned vo = (unsigned)((q % 3) * 64) * (E * 2) + (unsigned)(32 * j * 2) + wp_lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wp_rsrc, (lds_ptr_t)(ring + pos * SLOT + wv * 1024), 16, vo, 0, 0, 0);
    };
    // (in the qkv tail only waves 0-3 issue DMA, two of the eight 1 KiB pieces of a slot each: lines 16 w .. 16 w + 15)
    auto issue_wq = [&](int q, int pos, unsigned lane_off) {
        if (DBG & 8) return;
        const unsigned vo = wq_slot_offset(q, lane_off, q_rot);
        char* dst = ring + pos * SLOT + (wv & 3) * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wq_rsrc, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wq_rsrc, (lds_ptr_t)(dst + 1024), 16, vo == OOB ? OOB : vo + 8 * (E * 2), 0, 0, 0);
    };
    auto issue_rel = [&](int it, int g) {
        const int pos = (g + 12 + 24 + 48 + PRE) & (NSLOT - 1);  // PRE + 12 + 24 it + g, multiples of 8 dropped
        if (PROJ && g < -12) issue_wp(g + 48, pos);
        else if (g < 0) issue_w1(0, g + 12, pos);
        else if (g < 12) issue_w1(it + 1, g, pos);
        else if (g < 24) issue_w2(it, g - 12, pos);
        else issue_w1(it + 2, g - 24, pos);
    };

    // ---- fragment reads (ring positions as above)
    const int frag_sw = f_row & 7;
    const int rows0 = rg * 48 + f_row;
    auto opaque = [](u32x4& v) { asm volatile("" : "=v"(v)); };  // dev only: a fragment that costs no LDS read
    // phase A, step kt of a tile sequence starting at stream index g0: W1 fragments (2 units-of-16 x 2 k-halves) and
    // h fragments (3 row blocks x 2 k-halves)
    auto read_A = [&](int g0, int kt, u32x4 (&wf)[2][2], u32x4 (&hf)[3][2]) {
        if (DBG & 16) {
            for (int ks = 0; ks < 2; ++ks) { for (int nf = 0; nf < 2; ++nf) opaque(wf[nf][ks]); for (int rf = 0; rf < 3; ++rf) opaque(hf[rf][ks]); }
            return;
        }
        const char* wbase = ring + ((g0 + PRE + 2 * kt + (cg >> 1)) & (NSLOT - 1)) * SLOT + ((cg & 1) * 32 + f_row) * ROW_BYTES;
        const char* hbase = smem + OFF_HS + kt * HS_KB + rows0 * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + f_kg) ^ frag_sw) << 4;
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) wf[nf][ks] = *reinterpret_cast<const u32x4*>(wbase + nf * 16 * ROW_BYTES + ch);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) hf[rf][ks] = *reinterpret_cast<const u32x4*>(hbase + rf * 16 * ROW_BYTES + ch);
        }
    };
    // phase B, step j: W2 fragments (6 outputs-of-16) ...
    auto read_Bw = [&](int g0, int j, u32x4 (&wf)[6]) {
        if (DBG & 16) { for (int nf = 0; nf < 6; ++nf) opaque(wf[nf]); return; }
        const int ch = (((cg >> 1) * 4 + f_kg) ^ frag_sw) << 4;
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) {
            const int line0 = (cg & 1) * 96 + nf * 16;
            const int pos = (g0 + PRE + 3 * j + (line0 >> 6)) & (NSLOT - 1);
            wf[nf] = *reinterpret_cast<const u32x4*>(ring + pos * SLOT + ((line0 & 63) + f_row) * ROW_BYTES + ch);
        }
    };
    // ... and row-operand fragments (3 row blocks) of k-step j (32 wide) of the G tile or of the input rows
    auto read_rows = [&](int region, int j, u32x4 (&gf)[3]) {
        if (DBG & 16) { for (int rf = 0; rf < 3; ++rf) opaque(gf[rf]); return; }
        const char* gbase = smem + region + (j >> 1) * HS_KB + rows0 * ROW_BYTES + ((((j & 1) * 4 + f_kg) ^ frag_sw) << 4);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) gf[rf] = *reinterpret_cast<const u32x4*>(gbase + rf * 16 * ROW_BYTES);
    };

    // dev only: time stamp of step st of iteration it (block 0, waves 0 and 4)
    auto stamp = [&](int it, int st) {
        if (!(DBG & 512)) return;
        if (blockIdx.x != 0 || (wv & 3) != 0) return;
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) p.trace[(wv >> 2) * 4096 + it * 10 + st] = t;
    };

    stamp(20, 0);  // kernel start
    f32x4 acc[3][6];  // the 96 x 384 block this workgroup owns: residual + bias, then the projection / FFN accumulate on it
    bool valid[3];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) valid[rf] = m0 + rg * 48 + f_row + rf * 16 < p.M;
    if (ATT) {
        // ================= attention of this workgroup's 96 query rows. Sequence = 192 tokens = this workgroup's rows and its
        // neighbour's. The work is 12 heads x 6 query tiles of 16 = 72 (head, tile) tasks, dealt round-robin to the EIGHT waves
        // in nine rounds: task t = 8 round + wave -> head t / 6, tile t % 6, so every SIMD carries two waves in every round
        // (round 2 kept wave w on tile w for every head: six busy waves, two SIMDs with two tiles each and the phase as long
        // as those - 12 x 2.0 us; the softmax is VALU-bound, 1540 VALU cycles per task against 384 of MFMA). A round touches
        // two consecutive heads. K and V of a head (192 x 32 each, 12 KiB + 12 KiB) arrive by LDS-DMA, all eight waves issuing,
        // into FOUR buffers in rotation (head h -> buffer h & 3): two in the ring's region and one in the G region, which the
        // weight stream / the FFN only need afterwards, and one made of the last quarter of the ring (K) and the LAST k-block of
        // the row image (V): that block only receives the outputs of heads 10 and 11 (rounds 7, 8), the buffer's last tenant
        // is head 8 (round 6). Head h + 4 is requested when the last round that reads head h has ended, one or two rounds
        // before its own first round.
        // Both operands stay row-major by key:
        //   * K chunk (key, c) sits at 16-byte position 4 key + (c ^ ((key >> 2) & 3)) - the 16 keys x one chunk a
        //     ds_read_b128 of the S^T = K Q^T operand touches then hit 16 different bank groups;
        //   * V chunk (key, c) at 4 key + (c ^ 2 ((key >> 2) & 1)); the O^T = V^T P^T operand - eight keys of one head
        //     dimension per lane - comes out of two ds_read_b64_tr_b16, the gfx950 transposing read: within 16 lanes, lane
        //     4 r + q supplies the address of four consecutive 16-bit elements M[r][4 q ..], lane i receives M[0..3][i]
        //     (scripts/micro/tr_probe.hip). No register staging, no 16-bit scatter.
        constexpr int SEQ = 192, NKT = SEQ / 16, HEADS = E / 32, RS = 3 * E, NR = 9, TILES = BM / 16;
        constexpr int KBYTES = SEQ * 64;  // K [192][64 B] or V [192][64 B] of one head
        static_assert(HEADS * TILES == NR * WAVES, "72 tasks in nine rounds of eight");
        static_assert(4 * KBYTES + KBYTES <= NSLOT * SLOT && 2 * KBYTES <= 2 * HS_KB && KBYTES == HS_KB, "buffer map");
        const int srow0 = (m0 / SEQ) * SEQ;
        const __amdgpu_buffer_rsrc_t qkv_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.qkv_in), 0, (unsigned)p.M * (unsigned)(RS * 2), 0x00020000);
        // byte offsets inside the LDS allocation of the K and the V half of head hd's buffer
        auto k_off = [](int hd) -> int { const int b = hd & 3; return b == 0 ? OFF_RING + 4 * KBYTES : b == 1 ? OFF_RING : b == 2 ? OFF_RING + 2 * KBYTES : OFF_GS; };
        auto v_off = [](int hd) -> int { const int b = hd & 3; return b == 0 ? OFF_HS + 5 * HS_KB : b == 1 ? OFF_RING + KBYTES : b == 2 ? OFF_RING + 3 * KBYTES : OFF_GS + KBYTES; };
        // instruction i of a head (24 of 1 KiB): i < 12 K keys 16 i .., else V keys 16 (i - 12) ..; lane = (key, position)
        const int a_key = lane >> 2, a_pos = lane & 3;
        auto issue_head = [&](int hd) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = wv + 8 * u;
                const bool isv = i >= 12;
                const int key = 16 * (isv ? i - 12 : i) + a_key;
                const int c = isv ? (a_pos ^ (2 * ((key >> 2) & 1))) : (a_pos ^ ((key >> 2) & 3));
                const unsigned vo = (unsigned)(srow0 + key) * (unsigned)(RS * 2) + (unsigned)((isv ? 2 * E : E) * 2 + hd * 64 + c * 16);
                char* dst = smem + (isv ? v_off(hd) + (i - 12) * 1024 : k_off(hd) + i * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qkv_rsrc, (lds_ptr_t)dst, 16, vo, 0, 0, 0);  // rows past M: out of bounds = zeros
            }
        };
        // order of the first requests: K / V of heads 0 and 1 (round 0), the Q fragments of this wave's nine tasks, heads 2, 3
        issue_head(0);
        issue_head(1);
        u32x4 qf[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = 8 * r + wv, hd = t / TILES, tl = t - hd * TILES;
            const int qm = m0 + tl * 16 + f_row;
            qf[r] = *reinterpret_cast<const u32x4*>(p.qkv_in + (size_t)(qm < p.M ? qm : p.M - 1) * RS + hd * 32 + f_kg * 8);
        }
        issue_head(2);
        issue_head(3);
        const int k_frag_off = f_row * 64 + ((f_kg ^ ((f_row >> 2) & 3)) << 4);  // + kt * 1024
        // transposing V read: lane supplies key 4 f_kg + (f_row >> 2) of the 16-key tile, elements 4 (f_row & 3) .. + 3 of the
        // 16-dimension tile dt: chunk 2 dt + ((f_row & 3) >> 1), swizzled by 2 (f_kg & 1), upper or lower half
        // (chunk (2 dt + q) ^ 2 (f_kg & 1) = 2 (dt ^ (f_kg & 1)) + q)
        const int v_frag_off = (4 * f_kg + (f_row >> 2)) * 64 + (((f_row & 3) >> 1) << 4) + (f_row & 1) * 8;
        const int v_dt_off[2] = {(f_kg & 1) * 32, ((f_kg & 1) ^ 1) * 32};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            stamp(30, r);
            // The heads of round r have landed. Program order of this wave's vector-memory requests:
            //   h0 h1 | Q x9 | h2 h3 || r0: res | r1: h4 res | r2: h5 res | r3: h6 h7 res | r4: h8 res | r5: h9 res | r6: h10 h11 res |
            //   r7: res | r8: res + 3 weight slots      (a head = 3 requests per wave, res = 2 residual loads)
            // first rounds of the heads: 0 0 1 2 3 3 4 5 6 6 7 8; the wait of round r lets exactly the requests YOUNGER than the
            // last head it needs fly on ("at most as many outstanding as there are younger requests" is always safe)
            if (r == 0) wait_dma_and_barrier<NR + 6>();
            else if (r % 3 == 1) wait_dma_and_barrier<5>();                   // r1: h3 res0 - r4: h7 res3 - r7: h11 res6
            else if (r == 2 || r == 5) wait_dma_and_barrier<7>();             // r2: res0 h4 res1 - r5: res3 h8 res4
            else if (r == 8) wait_dma_and_barrier<4>();                       // res6 res7
            else wait_dma_and_barrier<2>();                                   // r3: res2 - r6: res5
            stamp(32, r);
            // past the barrier the heads whose last round was r - 1 are done with: their buffers take the heads four further on
            if (r == 1) issue_head(4);
            if (r == 2) issue_head(5);
            if (r == 3) { issue_head(6); issue_head(7); }
            if (r == 4) issue_head(8);
            if (r == 5) issue_head(9);
            if (r == 6) { issue_head(10); issue_head(11); }
            {
                // The residual rows (147 KB per workgroup, fp32) trickle in under the attention math, two loads per round:
                // plain loads into the accumulators - the bias is added after the phase, an add (or a select: rows past M
                // read row M - 1, nothing of them is ever stored) here would wait for them.
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = r * 2 + u, rf = i / 6, nf = i % 6;
                    const int m = m0 + rg * 48 + f_row + rf * 16;
                    acc[rf][nf] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(m < p.M ? m : p.M - 1) * E + cg * 96 + nf * 16 + f_kg * 4);
                }
            }
            if (r + 1 == NR) {
                // last round: heads 10 and 11 sit in the second ring buffer and in the G region; the first ring buffer (head 9,
                // rounds 6-7) is free, the first three slots of the weight stream - the projection's first tile - fly under it
#pragma unroll
                for (int q = 0; q < 3; ++q) issue_rel(0, q - 12 - PRE);
            }
            {
                const int t = 8 * r + wv, hd = t / TILES, tl = t - hd * TILES;  // wave-uniform
                const char* Ks = smem + k_off(hd);
                const char* Vs = smem + v_off(hd);
                const u32x4 qh = qf[r];
                f32x4 sc[NKT];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)  // S^T: lane holds keys 16 kt + 4 f_kg + (0..3) of query f_row
                    { u32x4 kf; if (ATT_DBG & 4) asm volatile("" : "=v"(kf)); else kf = *reinterpret_cast<const u32x4*>(Ks + kt * 1024 + k_frag_off); sc[kt] = mma(kf, qh, f32x4{0.f, 0.f, 0.f, 0.f}); }
                float mx = -__builtin_inff();
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sc[kt][i]);
                {
                    // max over the four lane groups of a query: the gfx950 row swaps (v_permlane16_swap / v_permlane32_swap,
                    // plain VALU) instead of two ds_bpermute round trips through the LDS queue
                    const unsigned mu = __builtin_bit_cast(unsigned, mx);
                    const auto s16 = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
                    mx = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
                    const unsigned mv = __builtin_bit_cast(unsigned, mx);
                    const auto s32 = __builtin_amdgcn_permlane32_swap(mv, mv, false, false);
                    mx = fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
                }
                const float mb = mx * p.scale_log2e;
                // The row sums come out of the matrix pipe: a third "V^T" fragment of all ones gives sum_k P[q][k] in every
                // row of its 16 x 16 block - 48 additions and two cross-lane hops per lane less on the VALU, which bounds this
                // phase; the pipe has the room (6 more MFMAs per task). It is the sum of the bf16-rounded weights, i.e. of
                // exactly the numbers the output is a combination of.
                f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                f32x4 osum = {0.f, 0.f, 0.f, 0.f};
                const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
                // O^T = V^T P^T; the key order inside a 32-key block is a permutation shared by both operands. The transposing
                // reads are issued as inline asm with their own counted lgkmcnt waits (LDS returns in order; whatever else the
                // compiler puts into that queue only makes a "<= N outstanding" wait stricter): as a builtin the read carries
                // no memory operand, and the compiler then holds it back with a vmcnt wait until every LDS-DMA in flight has
                // landed - including the K / V of FUTURE rounds requested a few hundred cycles earlier, whose whole point is to
                // fly under this math (round 2 stamps: 4 000 cycles per head against 2 x 1 500 of VALU work).
                const unsigned vaddr = (unsigned)(__SIZE_TYPE__)(lds_ptr_t)(Vs + v_frag_off);
                u32x2 vr[2][2][2];  // [buffer][dt][lo / hi]
                auto read_v = [&](int blk, u32x2 (&dst)[2][2]) {
                    if (ATT_DBG & 2) { for (int dt = 0; dt < 2; ++dt) for (int h = 0; h < 2; ++h) asm volatile("" : "=v"(dst[dt][h])); return; }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const unsigned ad = vaddr + (unsigned)(blk * 2048) + (unsigned)v_dt_off[dt];  // keys 32 blk + .., chunks 2 dt, 2 dt + 1 (swizzled)
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst[dt][0]) : "v"(ad));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(dst[dt][1]) : "v"(ad));
                    }
                };
                // exp of the scores of one 32-key block (arguments <= 0)
                auto exp_blk = [&](int blk) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            sc[2 * blk + kk][i] = (ATT_DBG & 1) ? __builtin_fmaf(sc[2 * blk + kk][i], p.scale_log2e, -mb) : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2 * blk + kk][i], p.scale_log2e, -mb));
                };
                // Software pipeline over the six key blocks: the V reads of block b + 1 are requested, then the exponentials of
                // block b + 1 run while they are in flight, then block b's MFMAs. All eight waves are in the same phase of a round
                // at the same time: as [all K reads | all exponentials | all V reads] the LDS pipe (192 KB per round = 1 536
                // cycles) and the VALU took turns idling.
                read_v(0, vr[0]);
                exp_blk(0);
#pragma unroll
                for (int blk = 0; blk < NKT / 2; ++blk) {
                    const int cur = blk & 1;
                    if (blk + 1 < NKT / 2) {
                        read_v(blk + 1, vr[cur ^ 1]);
                        exp_blk(blk + 1);
                        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vr[cur][0][0]), "+v"(vr[cur][0][1]), "+v"(vr[cur][1][0]), "+v"(vr[cur][1][1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[cur][0][0]), "+v"(vr[cur][0][1]), "+v"(vr[cur][1][0]), "+v"(vr[cur][1][1]));
                    }
                    const f32x4 p0 = sc[2 * blk], p1 = sc[2 * blk + 1];
                    const bf16x8 pf = {(__bf16)p0[0], (__bf16)p0[1], (__bf16)p0[2], (__bf16)p0[3],
                                       (__bf16)p1[0], (__bf16)p1[1], (__bf16)p1[2], (__bf16)p1[3]};
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const u32x4 vf = {vr[cur][dt][0][0], vr[cur][dt][0][1], vr[cur][dt][1][0], vr[cur][dt][1][1]};
                        if (ATT_DBG & 8) { o[dt][0] += __builtin_bit_cast(float, vf[0] ^ vf[1] ^ vf[2] ^ vf[3]) + (float)pf[0] + (float)pf[2] + (float)pf[4] + (float)pf[6]; continue; }
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
                    }
                    if (ATT_DBG & 8) osum[0] += 1.0f; else
                    osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, osum, 0, 0, 0);
                }
                // lane holds d = 16 dt + 4 f_kg + (0..3) of query f_row: 8 bytes into the row-operand image of the projection
                // (column n = 32 hd + 16 dt + 4 f_kg: k-block hd >> 1, 16-byte chunk 4 (hd & 1) + 2 dt + (f_kg >> 1))
                const float inv = 1.0f / osum[0];
                const int row = tl * 16 + f_row;
                char* orow = smem + OFF_HS + (hd >> 1) * HS_KB + row * ROW_BYTES + (f_kg & 1) * 8;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f32x4 v = o[dt] * inv;
                    const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(orow + ((((hd & 1) * 4 + dt * 2 + (f_kg >> 1)) ^ (f_row & 7)) << 4)) = ov;
                }
            }
        }
        stamp(30, NR);
        __syncthreads();  // the last round's math is done: the ring and the G region are free, the rows are in place
#pragma unroll
        for (int q = 3; q < NSLOT; ++q) issue_rel(0, q - 12 - PRE);  // the rest of the first eight weight slots
    }

    // ---- prologue: ring filled with the first eight slots, input rows into LDS, accumulators = residual + b2
    if (!ATT) {
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) issue_rel(0, q - 12 - PRE);
    }
    if (!ATT) {
        // 72 DMA instructions, nine per wave: instruction i covers k-block i / 12, rows 8 (i % 12) .. +7
#pragma unroll
        for (int jj = 0; jj < 9; ++jj) {
            const int i = wv * 9 + jj;
            const int kb = i / 12, row = (i % 12) * 8 + (lane >> 3);
            const unsigned vo = (unsigned)(m0 + row) * (E * 2) + (unsigned)(kb * 128) + (unsigned)(d_lc << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(h_rsrc, (lds_ptr_t)(smem + OFF_HS + i * 1024), 16,
                                                     (m0 + row) < p.M ? vo : OOB, 0, 0, 0);
        }
    }
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
        const int m = m0 + rows0 + rf * 16;
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) {
            const int n = cg * 96 + nf * 16 + f_kg * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>((PROJ ? p.bp : p.b2) + n);
            if (ATT) v += acc[rf][nf];  // (the residual came in under the attention phase)
            else if (valid[rf]) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * E + n);
            acc[rf][nf] = v;
        }
    }
    wait_dma_and_barrier<0>();

    u32x4 wa[2][2][2], ha[2][3][2];  // phase-A fragments, double-buffered
    u32x4 wb[2][6], gb[2][3];        // phase-B fragments, double-buffered
    f32x4 pacc[3][2];                // P of the chunk in phase A
    f32x4 pold[3][2];                // P of the previous chunk, on its way through GELU
    f32x4 b1v[2];                    // b1 of the chunk in pold
    auto load_b1 = [&](int ci) {
        if (DBG & 64) {  // dev: a register-only stand-in (an undefined b1v would let the compiler drop the GELU with it)
            const float t = (float)(ci + lane) * 1e-3f;
            b1v[0] = f32x4{t, t, t, t};
            b1v[1] = b1v[0];
            return;
        }
        const int c = chunk_of(ci < nchunks ? ci : 0);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) b1v[nf] = *reinterpret_cast<const f32x4*>(p.b1 + c * CHUNK + cg * 32 + nf * 16 + f_kg * 4);
    };
    // GELU of one accumulator fragment of pold -> bf16 -> operand tile of phase B. Lane holds units
    // 32 cg + 16 nf + 4 f_kg + (0..3) of its rows: k-block cg >> 1 of the chunk, 16-byte chunk
    // 4 (cg & 1) + 2 nf + (f_kg >> 1), upper or lower 8 bytes.
    auto gelu_frag = [&](int rf, int nf) {
        char* gs = smem + OFF_GS + (cg >> 1) * HS_KB + rows0 * ROW_BYTES + (f_kg & 1) * 8;
        const f32x4 v = pold[rf][nf] + b1v[nf];
        const bf16x4 g = (DBG & 2) ? bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]} : gelu4_bf16(v);
        const int ch = ((cg & 1) * 4 + 2 * nf + (f_kg >> 1)) ^ frag_sw;
        *reinterpret_cast<bf16x4*>(gs + rf * 16 * ROW_BYTES + (ch << 4)) = g;
    };
    // One phase-A step: tile kt is in registers (wa/ha[kt & 1]); read the next tile (or the first phase-B tile),
    // issue the DMA for the slots this step frees, run the MFMAs - and, in between, the GELU of one fragment of the
    // PREVIOUS chunk, whose VALU work hides under this chunk's MFMAs.
    // Issue order inside a step: the MFMAs only need fragments read one step ago, so they start at once and the next
    // tile's LDS reads trickle in between them - a burst of reads from all eight waves would hold every wave in the
    // LDS queue (in-order issue) while the matrix pipe idles.
    auto step_A = [&](int it, int g0, int kt, bool with_gelu, bool last_of_peeled) {
        const int cur = kt & 1;
        stamp(it + (g0 < 0 ? 0 : 1), kt);
        if (kt < KT1 - 1 || last_of_peeled) wait_dma_and_barrier<NSLOT - 2 - 2>();
        else wait_dma_and_barrier<NSLOT - 2 - 3>();
        issue_rel(it, g0 + 2 * kt + NSLOT);
        issue_rel(it, g0 + 2 * kt + NSLOT + 1);
        if (kt < KT1 - 1) read_A(g0 + 12, kt + 1, wa[cur ^ 1], ha[cur ^ 1]);
        else if (last_of_peeled) read_A(g0 + 12 + 12, 0, wa[0], ha[0]);
        else read_Bw(g0 + 12 + 12, 0, wb[0]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    if (!(DBG & 4)) pacc[rf][nf] = mma(wa[cur][nf][ks], ha[cur][rf][ks], pacc[rf][nf]);
                }
        if (with_gelu) gelu_frag(kt >> 1, kt & 1);
        // issue order: one MFMA, a few VALU (GELU), one LDS read, ... - VALU and LDS work sits in the MFMA shadows
        if (kt < KT1 - 1 || last_of_peeled) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                SGB(SG_MFMA, 1);
                if (with_gelu) SGB(SG_VALU, 4);
                if (i < 10) SGB(SG_DS_READ, 1);
                if (i == 1 || i == 5) SGB(SG_VMEM, 1);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                SGB(SG_MFMA, 1);
                if (with_gelu) SGB(SG_VALU, 4);
                if (i < 6) SGB(SG_DS_READ, 1);
                if (i == 1 || i == 5) SGB(SG_VMEM, 1);
            }
        }
    };

    // LayerNorm statistics of the accumulator rows: a row's 384 values sit in 4 lane groups x 4 column waves; `stat` is
    // 8 x 96 floats of LDS nobody else uses at that moment
    auto row_stats = [&](float* stat, float (&mean)[3], float (&rstd)[3]) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            float sm = 0.f;
#pragma unroll
            for (int nf = 0; nf < 6; ++nf) {
                const f32x4 v = acc[rf][nf];
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
            for (int nf = 0; nf < 6; ++nf)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = acc[rf][nf][k] - mean[rf];
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
    };

    stamp(20, 1);  // prologue done (rows, residual, first tiles)
    if (PROJ) {
        // ================= attention output projection: acc (= x + bp) += a Wp^T, twelve steps of k = 32 in the style
        // of phase B with the input rows as the row operand; then h = LayerNorm2(acc) replaces the rows in LDS.
        read_Bw(-PRE, 0, wb[0]);  // the projection tiles start the stream: ring position = slot index
        // (rot_mod(j, p_rot, 12): the k-step whose weights arrive at stream step j)
        read_rows(OFF_HS, rot_mod(0, p_rot, 12), gb[0]);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int cur = j & 1;
            if (j < 11) wait_dma_and_barrier<NSLOT - 3 - 3>();
            else wait_dma_and_barrier<NSLOT - 3 - 2>();
#pragma unroll
            for (int i = 0; i < 3; ++i) issue_rel(0, -48 + 3 * j + NSLOT + i);
            if (j < 11) {
                read_Bw(-PRE, j + 1, wb[cur ^ 1]);
                read_rows(OFF_HS, rot_mod(j + 1, p_rot, 12), gb[cur ^ 1]);
            }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) {
                    if (!(DBG & 4)) acc[rf][nf] = mma(wb[cur][nf], gb[cur][rf], acc[rf][nf]);
                }
            if (j < 11) {
#pragma unroll
                for (int i = 0; i < 9; ++i) { SGB(SG_MFMA, 2); SGB(SG_DS_READ, 1); if (i == 0 || i == 3 || i == 6) SGB(SG_VMEM, 1); }
            }
        }
        // every wave has passed the barrier of step 11, so nobody reads the input rows any more; G is not in use yet
        float mean2[3], rstd2[3];
        row_stats(reinterpret_cast<float*>(smem + OFF_GS), mean2, rstd2);
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) {
            const int n = cg * 96 + nf * 16 + f_kg * 4;  // k-block n >> 6, 16-byte chunk (n & 63) >> 3, 8-byte half f_kg & 1
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma2 + n), b = *reinterpret_cast<const f32x4*>(p.beta2 + n);
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(p.b2 + n);
            char* hs = smem + OFF_HS + (n >> 6) * HS_KB + rows0 * ROW_BYTES + (((((n & 63) >> 3)) ^ frag_sw) << 4) + (f_kg & 1) * 8;
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                const f32x4 v = acc[rf][nf];
                const float mu = mean2[rf], rs = rstd2[rf];
                const bf16x4 hv = {(__bf16)((v[0] - mu) * rs * g[0] + b[0]), (__bf16)((v[1] - mu) * rs * g[1] + b[1]),
                                   (__bf16)((v[2] - mu) * rs * g[2] + b[2]), (__bf16)((v[3] - mu) * rs * g[3] + b[3])};
                *reinterpret_cast<bf16x4*>(hs + rf * 16 * ROW_BYTES) = hv;
                acc[rf][nf] = v + b2v;  // the FFN accumulates on x' + b2
            }
        }
        wait_dma_and_barrier<0>();  // h complete, first phase-A tiles landed
    }

    stamp(20, 2);  // projection + ln2 done
    // ---- peeled phase A of the first chunk
    read_A(0, 0, wa[0], ha[0]);
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) step_A(0, -12, kt, false, kt == KT1 - 1);
    load_b1(0);

    for (int it = 0; it < nchunks; ++it) {
        // ================= phase A of chunk it + 1 (P = h W1[chunk]^T, 48 rows x 32 units) over GELU of chunk it
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                pold[rf][nf] = pacc[rf][nf];
                pacc[rf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) step_A(it, 0, kt, true, false);

        // ================= phase B of chunk it: acc[48 rows x 96 outputs] += G W2[:, chunk]^T
#pragma unroll
        for (int j = 0; j < KT2; ++j) {
            const int cur = j & 1;
            stamp(it + 1, KT1 + j);
            if (j < KT2 - 1) wait_dma_and_barrier<NSLOT - 3 - 3>();
            else wait_dma_and_barrier<NSLOT - 3 - 2>();
#pragma unroll
            for (int i = 0; i < 3; ++i) issue_rel(it, 12 + 3 * j + NSLOT + i);
            if (j == 0) read_rows(OFF_GS, 0, gb[0]);  // G exists only after the barrier above
            if (j < KT2 - 1) {
                read_Bw(12 + 12, j + 1, wb[cur ^ 1]);
                read_rows(OFF_GS, j + 1, gb[cur ^ 1]);
            } else {
                read_A(12 + 24, 0, wa[0], ha[0]);  // first phase-A tile of the next iteration
                load_b1(it + 1);
            }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) {
                    if (!(DBG & 4)) acc[rf][nf] = mma(wb[cur][nf], gb[cur][rf], acc[rf][nf]);
                }
            if (j == 0) SGB(SG_DS_READ, 3);
            if (j < KT2 - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) { SGB(SG_MFMA, 2); SGB(SG_DS_READ, 1); if (i == 0 || i == 3 || i == 6) SGB(SG_VMEM, 1); }
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) { if (i < 2) SGB(SG_MFMA, 1); else SGB(SG_MFMA, 2); SGB(SG_DS_READ, 1); if (i == 0 || i == 3 || i == 6) SGB(SG_VMEM, 1); }
            }
        }
    }

    stamp(20, 3);  // FFN loop done
    // ---- LayerNorm epilogue (G is out of use: its region carries the statistics exchange while the ring may already
    // hold qkv tiles)
    wait_dma_and_barrier<63>();
    float mean[3], rstd[3];
    row_stats(reinterpret_cast<float*>(smem + OFF_GS), mean, rstd);
    stamp(21, 0);  // row statistics done
    // Everything lane-dependent below is recomputed from a laundered lane id: these addresses are loop-invariant, the
    // compiler would hoist them above the FFN loop, and that loop has no register to spare (the same trick as in the
    // epilogue of pp_gemm.hip).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int e_row = lane_e & 15, e_kg = lane_e >> 4, e_rows0 = rg * 48 + e_row, e_sw = e_row & 7;
    unsigned wq_lane = 0;
    if (QKV) {
        const int lc = (lane_e & 7) ^ (lane_e >> 3);
        wq_lane = (unsigned)((wv & 3) * 16 + (lane_e >> 3)) * (E * 2) + (unsigned)(lc << 4);  // W1-style lines, two pieces per wave
        // the statistics exchange ended with every DMA of the FFN landed (the last ones were out-of-bounds fillers), so
        // the ring restarts at slot 0 with the first qkv tiles; they fly under the stores below
        if (rg == 0) {
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) issue_wq(q, q, wq_lane);
        }
    }
#pragma unroll
    for (int nf = 0; nf < 6; ++nf) {
        const int n = cg * 96 + nf * 16 + e_kg * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + n), b = *reinterpret_cast<const f32x4*>(p.beta + n);
        char* hs = smem + OFF_HS + (n >> 6) * HS_KB + e_rows0 * ROW_BYTES + (((((n & 63) >> 3)) ^ e_sw) << 4) + (e_kg & 1) * 8;
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            const int m = m0 + e_rows0 + rf * 16;
            const size_t off = (size_t)m * E + n;
            const f32x4 v = acc[rf][nf];
            const float mu = mean[rf], rs = rstd[rf];
            const bf16x4 hv = {(__bf16)((v[0] - mu) * rs * g[0] + b[0]), (__bf16)((v[1] - mu) * rs * g[1] + b[1]),
                               (__bf16)((v[2] - mu) * rs * g[2] + b[2]), (__bf16)((v[3] - mu) * rs * g[3] + b[3])};
            if (QKV) *reinterpret_cast<bf16x4*>(hs + rf * 16 * ROW_BYTES) = hv;  // the row operand of the qkv steps
            if (m >= p.M) continue;
            // (QKV: rows 48-95 - waves 4-7, which never count vmcnt in the tail - keep their x in registers across
            // the barrier below and store it at the head of the tail, see there)
            if (!QKV || rg == 0) *reinterpret_cast<f32x4*>(p.x_out + off) = v;
            if (p.h_out) *reinterpret_cast<bf16x4*>(p.h_out + off) = hv;
        }
    }
    stamp(20, 4);  // LayerNorm + x_out / h stores issued
    if (!QKV) {
        wait_dma_and_barrier<0>();  // the out-of-bounds DMAs past the last chunk must not outlive the workgroup
        return;
    }

    // ================= qkv of the next layer: six column blocks of 192 outputs x six k-steps of 64 (18 MFMAs per wave
    // and step, wave tile 48 rows x 48 outputs). Two accumulator sets: while block b+1 accumulates, block b leaves -
    // bias, bf16, one 48-row half at a time through the G region (512 B pitch, chunks XOR-swizzled by row & 7) so that
    // every row goes out as three whole cache lines - spread over four of the next block's steps instead of
    // one burst per block that stalls the whole chip on the HBM write queue. Roles by wave, because global stores share
    // vmcnt with the DMA stream and a counted wait would sit behind them: waves 0-3 issue all the DMA (and count it),
    // waves 4-7 do all the stores (and never wait for them before the end).
    wait_dma_and_barrier<0>();
    if (rg == 1) {
        // The 37.7 MB of fp32 residual rows leave at the ~4.2 TB/s the chip sustains on writes, and the barrier above waits for
        // them (the DMA-issuing waves 0-3 must drain their stores before their counted waits begin): 9 us per launch with
        // no MFMA running. Waves 4-7 never wait on vmcnt in the tail, so THEIR half goes out here, after the barrier, under the
        // first column block of the tail (whose own qkv stores only begin with the second block): the wait above then
        // covers half the bytes. The accumulators are still intact (the tail zeroes its first set after this).
#pragma unroll
        for (int nf = 0; nf < 6; ++nf) {
            const int n = cg * 96 + nf * 16 + e_kg * 4;
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                const int m = m0 + e_rows0 + rf * 16;
                if (m < p.M) *reinterpret_cast<f32x4*>(p.x_out + (size_t)m * E + n) = acc[rf][nf];
            }
        }
    }
    u32x4 wq_f[2][3][2], hq_f[2][3][2];  // fragments of the qkv steps, double-buffered: [buffer][fragment][k-half]
    auto read_Q = [&](int t, u32x4 (&wf)[3][2], u32x4 (&hf)[3][2]) {
        if (DBG & 16) {
            for (int ks = 0; ks < 2; ++ks) for (int f = 0; f < 3; ++f) { opaque(wf[f][ks]); opaque(hf[f][ks]); }
            return;
        }
        const char* hbase = smem + OFF_HS + (t % 6) * HS_KB + rows0 * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + f_kg) ^ frag_sw) << 4;
#pragma unroll
            for (int nfr = 0; nfr < 3; ++nfr) {
                const int line = cg * 48 + nfr * 16;  // first weight row of the fragment inside the 192-row tile (wave-uniform)
                const int pos = (3 * t + (line >> 6)) & (NSLOT - 1);
                wf[nfr][ks] = *reinterpret_cast<const u32x4*>(ring + pos * SLOT + ((line & 63) + f_row) * ROW_BYTES + ch);
            }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) hf[rf][ks] = *reinterpret_cast<const u32x4*>(hbase + rf * 16 * ROW_BYTES + ch);
        }
    };
    char* gst = smem + OFF_GS;
    // Row half `ps` (48 rows x 192 columns = three whole 128-byte lines per row) of the block held in accumulator set
    // `set` into the staging rows (512-byte pitch): the four waves that own those rows. The bias of the block was loaded
    // during the block itself (see the step loop for where, relative to the counted DMA waits).
    f32x4 bq_f[3];
    auto load_bias = [&](int bb) {
#pragma unroll
        for (int nfr = 0; nfr < 3; ++nfr) bq_f[nfr] = *reinterpret_cast<const f32x4*>(p.bq + rot_mod(bb, q_rot, 6) * 192 + cg * 48 + nfr * 16 + e_kg * 4);
    };
    auto stage_half = [&](int ps, int set) {
        if (rg != ps) return;
#pragma unroll
        for (int nfr = 0; nfr < 3; ++nfr) {
            const int byte = (cg * 48 + nfr * 16 + e_kg * 4) * 2;
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                const f32x4 v = acc[rf][set * 3 + nfr] + bq_f[nfr];
                const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                const int row = rf * 16 + e_row;  // row inside the half
                *reinterpret_cast<bf16x4*>(gst + row * 512 + ((((byte >> 4) ^ (row & 7)) << 4) | (byte & 15))) = ov;
            }
        }
    };
    // the staged half to HBM: waves 4-7, 48 rows x 24 chunks of 16 bytes
    auto store_half = [&](int bb, int ps, int i_lo, int i_hi) {
        if (rg == 0) return;
#pragma unroll
        for (int i5 = i_lo; i5 < i_hi; ++i5) {
            const int idx = i5 * 256 + (tid - 256);  // (tid is lane-dependent: fine, used only here)
            const int row = idx / 24, ch = idx - row * 24;
            if (idx < 48 * 24) {
                const int m = m0 + ps * 48 + row;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(gst + row * 512 + ((ch ^ (row & 7)) << 4));
                if (m < p.M && !(DBG & 1024))
                    *reinterpret_cast<u32x4*>(p.qkv + (size_t)m * (3 * E) + rot_mod(bb, q_rot, 6) * 192 + ch * 8) = raw;
            }
        }
    };
    read_Q(0, wq_f[0], hq_f[0]);
#pragma unroll
    for (int b = 0; b < 7; ++b) {  // block 6 is the drain: no MFMAs, only block 5 leaving
        const int set = b & 1;
        if (b < 6 && !(DBG & 2048)) {  // (ablation 2048: keep accumulating so that no MFMA of the tail is dead code)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nfr = 0; nfr < 3; ++nfr) acc[rf][set * 3 + nfr] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kt = 0; kt < (b < 6 ? 6 : 5); ++kt) {
            const int t = b * 6 + kt, cur = t & 1;
            stamp(22 + 2 * (b >> 1), (b & 1) * 6 + kt);
            if (rg == 0) wait_dma_and_barrier<2 * (NSLOT - 3 - 3)>();
            else wait_dma_and_barrier<63>();
            // this block's bias, for its own departure during the next block (the previous block's was last used at step
            // 3). Issued between the counted wait and the DMA of this step: older than that DMA, so the next counted
            // wait covers it for free.
            if (b < 6 && kt == 4) load_bias(b);
            if (rg == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) issue_wq(3 * t + NSLOT + i, (3 * t + NSLOT + i) & (NSLOT - 1), wq_lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (b < 6) {
                read_Q(t + 1, wq_f[cur ^ 1], hq_f[cur ^ 1]);  // (past the last step: a harmless read of the zero filler)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                        for (int nfr = 0; nfr < 3; ++nfr) {
                            if (!(DBG & 4)) acc[rf][set * 3 + nfr] = mma(wq_f[cur][nfr][ks], hq_f[cur][rf][ks], acc[rf][set * 3 + nfr]);
                        }
            }
            // The previous block leaves in the shadow of this block's MFMAs (which were issued first): rows 0-47 are
            // staged at step 0 and stored over steps 1-2, rows 48-95 staged at step 3 and stored over steps 4-5 (the
            // drain block has five steps: its second half goes out in one piece).
            if (b > 0 && !(DBG & 2048)) {
                if (kt == 0) stage_half(0, set ^ 1);
                if (kt == 1) store_half(b - 1, 0, 0, 3);
                if (kt == 2) store_half(b - 1, 0, 3, 5);
                if (kt == 3) stage_half(1, set ^ 1);
                if (kt == 4) store_half(b - 1, 1, 0, b < 6 ? 3 : 5);
                if (kt == 5) store_half(b - 1, 1, 3, 5);
            }
        }
    }
    if (DBG & 2048) {  // one live use of every accumulator
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 6; ++nf) t += acc[rf][nf];
        if (t[0] + t[1] + t[2] + t[3] == 123.456f) p.qkv[tid] = (__bf16)t[0];
    }
    stamp(20, 5);  // qkv tail done
    wait_dma_and_barrier<0>();  // the out-of-bounds DMAs past the last tile must not outlive the workgroup
    stamp(20, 6);
}

unsigned long long* g_trace = nullptr;
}  // namespace mlp
}  // namespace pp

#if MLP_DBG & 512
extern "C" void pp_mlp_set_trace(void* buf) { pp::mlp::g_trace = reinterpret_cast<unsigned long long*>(buf); }
#endif

namespace pp {
namespace mlp {
static int launch(const Params& p, bool proj, bool qkv, bool att, hipStream_t stream) {
    typedef void (*kern_t)(const Params);
    kern_t kern = static_cast<kern_t>(mlp_res_ln_kernel<false, false, false>);
    if (att && qkv) kern = static_cast<kern_t>(mlp_res_ln_kernel<true, true, true>);
    else if (att) kern = static_cast<kern_t>(mlp_res_ln_kernel<true, false, true>);
    else if (proj && qkv) kern = static_cast<kern_t>(mlp_res_ln_kernel<true, true, false>);
    else if (proj) kern = static_cast<kern_t>(mlp_res_ln_kernel<true, false, false>);
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(kern, dim3((p.M + BM - 1) / BM), dim3(THREADS), LDS, stream, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
}  // namespace mlp
}  // namespace pp

extern "C" int pp_mlp_residual_layernorm(const void* h_in, const void* w1, const float* b1, const void* w2,
                                         const float* b2, const float* residual, float* x_out, const float* gamma,
                                         const float* beta, float eps, void* h_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(h_in && w1 && b1 && w2 && b2 && residual && x_out && gamma && beta && h_out, PP_ERR_INVALID_ARG,
               "pp_mlp_residual_layernorm: NULL argument");
    PP_REQUIRE(E == mlp::E, PP_ERR_UNSUPPORTED, "pp_mlp_residual_layernorm: built for embed dim 384 (ViT-S)");
    PP_REQUIRE(M > 0 && F > 0 && F % mlp::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_mlp_residual_layernorm: hidden width must be a positive multiple of 128");
    PP_REQUIRE((size_t)F * E * 2 < 0x7ffffff0u && (size_t)M * E * 2 < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_mlp_residual_layernorm: operand exceeds 2 GiB");
    mlp::Params p{};
    p.h = reinterpret_cast<const __bf16*>(h_in);
    p.W1 = reinterpret_cast<const __bf16*>(w1);
    p.b1 = b1;
    p.W2 = reinterpret_cast<const __bf16*>(w2);
    p.b2 = b2;
    p.residual = residual;
    p.x_out = x_out;
    p.gamma = gamma;
    p.beta = beta;
    p.h_out = reinterpret_cast<__bf16*>(h_out);
    p.M = M;
    p.F = F;
    p.h_bytes = (unsigned)((size_t)M * E * 2);
    p.w1_bytes = (unsigned)((size_t)F * E * 2);
    p.w2_bytes = (unsigned)((size_t)E * F * 2);
    p.eps = eps;
    p.trace = mlp::g_trace;
    return mlp::launch(p, false, false, false, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_proj_mlp_residual_layernorm(const void* attn, const void* wp, const float* bp, const float* residual,
                                              const float* gamma2, const float* beta2, const void* w1, const float* b1,
                                              const void* w2, const float* b2, float* x_out, const float* gamma,
                                              const float* beta, float eps, void* h_out, const void* wqkv,
                                              const float* bqkv, void* qkv_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(attn && wp && bp && residual && gamma2 && beta2 && w1 && b1 && w2 && b2 && x_out && gamma && beta,
               PP_ERR_INVALID_ARG, "pp_proj_mlp_residual_layernorm: NULL argument");
    const bool qkv = wqkv != nullptr;
    PP_REQUIRE(qkv ? (bqkv && qkv_out) : (h_out != nullptr), PP_ERR_INVALID_ARG,
               "pp_proj_mlp_residual_layernorm: needs h_out, or wqkv + bqkv + qkv_out");
    PP_REQUIRE(E == mlp::E, PP_ERR_UNSUPPORTED, "pp_proj_mlp_residual_layernorm: built for embed dim 384 (ViT-S)");
    PP_REQUIRE(M > 0 && F > 0 && F % mlp::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_proj_mlp_residual_layernorm: hidden width must be a positive multiple of 128");
    PP_REQUIRE((size_t)F * E * 2 < 0x7ffffff0u && (size_t)M * E * 2 < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_proj_mlp_residual_layernorm: operand exceeds 2 GiB");
    mlp::Params p{};
    p.h = reinterpret_cast<const __bf16*>(attn);
    p.Wp = reinterpret_cast<const __bf16*>(wp);
    p.bp = bp;
    p.gamma2 = gamma2;
    p.beta2 = beta2;
    p.W1 = reinterpret_cast<const __bf16*>(w1);
    p.b1 = b1;
    p.W2 = reinterpret_cast<const __bf16*>(w2);
    p.b2 = b2;
    p.residual = residual;
    p.x_out = x_out;
    p.gamma = gamma;
    p.beta = beta;
    p.h_out = reinterpret_cast<__bf16*>(h_out);
    p.M = M;
    p.F = F;
    p.h_bytes = (unsigned)((size_t)M * E * 2);
    p.w1_bytes = (unsigned)((size_t)F * E * 2);
    p.w2_bytes = (unsigned)((size_t)E * F * 2);
    p.wp_bytes = (unsigned)((size_t)E * E * 2);
    p.Wq = reinterpret_cast<const __bf16*>(wqkv);
    p.bq = bqkv;
    p.qkv = reinterpret_cast<__bf16*>(qkv_out);
    p.wq_bytes = (unsigned)((size_t)3 * E * E * 2);
    p.eps = eps;
    p.trace = mlp::g_trace;
    return mlp::launch(p, true, qkv, false, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_vit_layer(const void* qkv_in, int seq_len, int heads, float scale, const void* wp, const float* bp,
                            const float* residual, const float* gamma2, const float* beta2, const void* w1,
                            const float* b1, const void* w2, const float* b2, float* x_out, const float* gamma,
                            const float* beta, float eps, void* h_out, const void* wqkv, const float* bqkv,
                            void* qkv_out, int M, int E, int F, void* stream) {
    using namespace pp;
    PP_REQUIRE(qkv_in && wp && bp && residual && gamma2 && beta2 && w1 && b1 && w2 && b2 && x_out && gamma && beta,
               PP_ERR_INVALID_ARG, "pp_vit_layer: NULL argument");
    const bool qkv = wqkv != nullptr;
    PP_REQUIRE(qkv ? (bqkv && qkv_out) : (h_out != nullptr), PP_ERR_INVALID_ARG, "pp_vit_layer: needs h_out, or wqkv + bqkv + qkv_out");
    PP_REQUIRE(qkv_out != qkv_in, PP_ERR_INVALID_ARG, "pp_vit_layer: qkv_out must not alias qkv_in (other workgroups still read it)");
    PP_REQUIRE(E == mlp::E && heads * 32 == E && seq_len == 192, PP_ERR_UNSUPPORTED,
               "pp_vit_layer: built for ViT-S at 256x192 (embed dim 384, 12 heads x 32, 192 tokens)");
    PP_REQUIRE(M > 0 && M % seq_len == 0 && F > 0 && F % mlp::CHUNK == 0, PP_ERR_UNSUPPORTED,
               "pp_vit_layer: M must be a multiple of the sequence length, the hidden width a multiple of 128");
    PP_REQUIRE((size_t)F * E * 2 < 0x7ffffff0u && (size_t)M * 3 * E * 2 < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_vit_layer: operand exceeds 2 GiB");
    mlp::Params p{};
    p.qkv_in = reinterpret_cast<const __bf16*>(qkv_in);
    p.scale_log2e = scale * 1.44269504088896340736f;
    p.h = reinterpret_cast<const __bf16*>(qkv_in);  // (unused: no row DMA in this mode)
    p.Wp = reinterpret_cast<const __bf16*>(wp);
    p.bp = bp;
    p.gamma2 = gamma2;
    p.beta2 = beta2;
    p.W1 = reinterpret_cast<const __bf16*>(w1);
    p.b1 = b1;
    p.W2 = reinterpret_cast<const __bf16*>(w2);
    p.b2 = b2;
    p.residual = residual;
    p.x_out = x_out;
    p.gamma = gamma;
    p.beta = beta;
    p.h_out = reinterpret_cast<__bf16*>(h_out);
    p.M = M;
    p.F = F;
    p.h_bytes = (unsigned)((size_t)M * E * 2);
    p.w1_bytes = (unsigned)((size_t)F * E * 2);
    p.w2_bytes = (unsigned)((size_t)E * F * 2);
    p.wp_bytes = (unsigned)((size_t)E * E * 2);
    p.Wq = reinterpret_cast<const __bf16*>(wqkv);
    p.bq = bqkv;
    p.qkv = reinterpret_cast<__bf16*>(qkv_out);
    p.wq_bytes = (unsigned)((size_t)3 * E * E * 2);
    p.eps = eps;
    p.trace = mlp::g_trace;
    return mlp::launch(p, true, qkv, true, reinterpret_cast<hipStream_t>(stream));
}
