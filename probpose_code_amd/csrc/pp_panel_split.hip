// Wide-tile GEMM for PP_PREC_F16X3 (split-fp16 operands, pp_split.h; the precision mode that meets the 1e-3 tolerance) on
// gfx950: its dense layers with long output rows - qkv and fc1 Linear layers, the two deconvolutions, the first tower convolution.
//
// In this format an operand element costs 4 bytes in LDS and HBM but an algorithmic product costs THREE fp16 MFMAs, so
// per byte staged the matrix pipe has 3x the work of the bf16 path: the 128 x 128 tiles of pp_gemm.hip (32 KiB per
// K-step per workgroup) leave it waiting for the L2 -> LDS fill. Here a workgroup owns the CU and a 192 x 256 or
// 256 x 192 tile - the layout of pp_panel_gemm.hip, which this file follows:
//
//   * 512 threads = 8 waves = two per SIMD, wave (rg, cg): row half rg, column quarter cg, RF x CF fragments of 16 x 16
//     (6 x 4 or 8 x 3), fp32 accumulators;
//   * a stage is ONE 128-byte block of K per row (32 elements: 32 hi halves | 32 lo halves), BM activation rows + BN
//     weight rows = 56 KiB, XOR-swizzled, filled by LDS-DMA; two stages; persistent workgroups, the stream of stages
//     runs on across output tiles;
//   * per stage 72 MFMAs per wave in two halves around ONE barrier:
//       first half   acc += Wh Ah          while the lo fragments (chunks 4..7) are read
//       -- barrier: every wave holds the stage in registers, its buffer takes the DMA of stage s + 2; stage s + 1 has landed
//       second half  acc += Wl Ah, then acc += Wh Al; the hi fragments of stage s + 1 are read into the registers of
//                    Ah / Wh as those die (row fragments during the first product, column fragments during the second);
//   * implicit im2col for the convolutions (a DMA lane computes its pixel / tap address, out-of-image taps are
//     out-of-bounds offsets = zeros), plain rows for Linear layers;
//   * epilogue: + bias, GELU (pp_split.h: erfc form, 1.5e-7) / ReLU, then a quarter of the tile's rows at a time as fp32 through a 48 KiB
//     staging region and out as the split format (8 elements per lane: 16 bytes of hi halves + 16 bytes of lo halves),
//     deconvolutions to the phase-interleaved output pixel.
//
// SPLIT = false instantiates the same kernel for bf16 operands (a stage is 64 elements of K, the two halves are its two
// 32-element k-blocks, one MFMA per fragment pair) for the long-K Linear layers with an fp32 residual epilogue that
// pp_panel_gemm.hip (convolutions, bf16 out) does not cover: fc2 of ViT-B (K = 3072) ran at ~550 TFLOP/s on the 128 x 128
// tiles. Short-K layers stay there (12 K-steps do not amortise this kernel's un-overlapped epilogue).
#include "pp_common.h"
#include "pp_gemm.h"
#include "pp_split.h"

#include <cstdlib>

namespace pp {
namespace psplit {

#ifndef PSPLIT_DBG
#define PSPLIT_DBG 0
#endif
// dev ablation switches (scripts/micro/psplit_ablate.sh), 0 in the product build: 1 no bias loads, 2 no residual loads,
// 128 no epilogue at all
constexpr int DBG = PSPLIT_DBG;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int THREADS = 512;
constexpr int CST_BYTES = 48 * 1024;  // epilogue staging of the two-stage form
// LDS of one instantiation: NST stages of (BM + BN) rows x 128 B; the two-stage form adds a separate staging region, the
// three-stage form stages the epilogue through the buffer of the stage it has just consumed
constexpr int lds_bytes(int BM, int BN, int NST) { return NST * (BM + BN) * 128 + (NST == 2 ? CST_BYTES : 0); }
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned OOB = 0x7ffffff0u;

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

template <bool SPLIT>
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if constexpr (SPLIT)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// NST = 3 (round 2): with two stages exactly ONE stage is in flight - the DMA of stage s + 1 is issued at the barrier of
// stage s - 1 and must have landed at the barrier of stage s, so a stage can never be shorter than the L2 -> LDS latency
// plus the transfer of its own bytes (measured 1.46 us per 56 KiB stage in the bf16 convolutions against 0.73 us of MFMA
// issue). Three stages keep two in flight (wait vmcnt(NI) instead of vmcnt(0)); they only fit with smaller stages
// (192 x 192 / 128 x 256 tiles: 48 KiB), and the epilogue staging then borrows the buffer of the last stage consumed -
// its refill is deferred until the epilogue is through.
// HEAD (round 3; split-fp16 deconvolution with N == BN == 256 only): the 1x1 convolution behind the last deconvolution
// (probmap_head.py:244-249) runs in the epilogue - ReLU'd accumulators are split in registers and ARE the MFMA operand (the
// contraction index of a K = 32 block is taken in the order the accumulator layout provides: channels 4 fg + i of two
// neighbouring 16-channel fragments; the 1x1 weights come pre-permuted to match, weights.py), every wave contracts its 64
// channels, the four column-group waves' partial sums meet in LDS (<= 28 maps x 96 pixels per pass), and the tile leaves as
// phase-separated fp32 logits: 13 KB instead of 196 KB per tile, the 403 MB feature map of bs 64 is neither written nor read
// back, the 1x1 launch disappears.
// POOL (round 3; split-fp16 3x3 convolution on 16 x 12 maps, 192 x 192 three-stage form: a tile = one image x 192 channels): the
// stage's MaxPool2d(4, 3) + ReLU (probmap_head.py:261-294: Conv - BN - MaxPool - ReLU) runs on the staged quarter - a quarter
// is four image rows = one row of pooling windows - and only the pooled (4, 4) map leaves: 12.6 MB instead of 151 MB at bs 64,
// the pooling launch and its re-read disappear.
template <int GATHER, int RF, int CF, bool SPLIT, int NST, bool HEAD = false, bool POOL = false>
__global__ __launch_bounds__(THREADS, 2) void panel_split_kernel(const GemmParams p) {
    constexpr int ESZ = SPLIT ? 4 : 2;   // bytes per operand element
    constexpr int KS = 128 / ESZ;        // elements of K per stage: one 128-byte line per row
    constexpr int BM = 32 * RF, BN = 64 * CF;
    constexpr int NA = BM / 8, JA = NA / 8;  // DMA instructions of the activation tile (8 lines each); per wave j < JA
    constexpr int STAGE = (BM + BN) * 128;   // bytes per stage
    constexpr int NI = (BM + BN) / 64;       // DMA instructions per wave and stage
    constexpr int OFF_CST = NST * STAGE;
    static_assert(NST == 2 || NST == 3, "two or three stages");
    static_assert((BM / 4) * BN * 4 <= (NST == 2 ? CST_BYTES : STAGE), "a quarter of the fp32 tile fits the staging region");
    static_assert(lds_bytes(BM, BN, NST) <= 160 * 1024, "LDS");
    static_assert(NA % 8 == 0 && JA <= 4 && RF % 2 == 0 && (BM + BN) % 64 == 0, "instruction split");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;

    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntn * ntm * p.groups;
    const int nsteps = p.K / KS;  // K = the length of ONE K-slice under split-K
    const int cps = p.Cin / KS;   // stages per tap (Linear: Cin = K, one "tap")
    // Split-K (3x3 convolutions with few output rows - the 4 x 4 stage of the scalar towers: 64 tiles for 256 CUs): `groups` counts
    // (K slice, problem) pairs, z = slice * nprob + problem. A slice is a CHANNEL range - channels [slice Cin / ksplit, ...) of all
    // nine taps, walked taps-inner like the unsplit convolution - so that any slice count dividing Cin / 32 works (four quarters of
    // 384 channels = 256 workgroups x 27 stages: one round of the chip); every slice writes its fp32 partial sums to its own plane
    // C + z strideC_z, the pooling kernel adds them up (pp_sum_maxpool_relu_nhwc).
    const int nprob = p.ksplit > 1 ? p.groups / p.ksplit : p.groups;
    const int cslice = p.ksplit > 1 ? p.Cin / p.ksplit : p.Cin;  // channels per slice

    // XCD-aware tile order as in pp_panel_gemm.hip: row panel -> group -> column tile, contiguous runs per XCD
    auto decode_tile = [&](int t, int& z, int& m0, int& n0) __attribute__((always_inline)) {
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        if (p.tile_order == 1) {
            // weight-major: a contiguous run of tiles (= what one XCD works on at a time) shares ONE (group, column tile) weight set
            // and walks the row panels - for the 3x3 convolutions of the split-fp16 mode, whose weight sets (2.65 MB per 192
            // columns) no longer fit the 4 MB L2 eight at a time
            m0 = (t % ntm) * BM;
            const int r = t / ntm;
            n0 = (r % ntn) * BN;
            z = r / ntn;
            return;
        }
        n0 = (t % ntn) * BN;
        const int r = t / ntn;
        z = r % p.groups;
        m0 = (r / p.groups) * BM;
    };

    // ---- DMA cursor (tile, stage), two stages ahead of the MFMAs
    const int d_l = lane >> 3;
    const unsigned d_kbytes = (unsigned)(((lane & 7) ^ d_l) << 4);
    unsigned a_voff[JA];
    int a_y[JA], a_x[JA];
    unsigned w_voff;
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    int i_tile = blockIdx.x, i_step = 0, i_tap = 0, i_c0 = 0, i_py = 0, i_px = 0, i_cbeg = 0;
    bool i_live = true;
    // (always_inline on every lambda that touches the per-instruction cursor arrays: called from several sites the compiler kept
    // setup_issue_tile / issue_instr out of line, which forces a_y / a_x / a_voff - captured by reference - into SCRATCH memory:
    // 7 - 9 scratch loads and stores per K-step inside the stage loop of every instantiation, rounds 2 - 3)
    auto setup_issue_tile = [&]() __attribute__((always_inline)) {
        int z = 0, m0 = 0, n0 = 0;
        i_live = i_tile < ntiles;
        if (i_live) decode_tile(i_tile, z, m0, n0);
        i_cbeg = 0;
        if (p.ksplit > 1) {  // (z of the output plane stays slice * nprob + problem: see the epilogue)
            i_cbeg = (z / nprob) * cslice;
            z = z % nprob;
        }
        if (GATHER == G_DECONV) {
            i_py = p.py < 0 ? (z >> 1) : p.py;
            i_px = p.py < 0 ? (z & 1) : p.px;
        }
        const char* Act = reinterpret_cast<const char*>(p.A) + (size_t)z * p.strideA_z * ESZ;
        const char* Wt = reinterpret_cast<const char*>(p.W) + (size_t)z * p.strideW_z * ESZ;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Act), 0, p.a_bytes, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, p.w_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int m = m0 + 8 * (wv + 8 * j) + d_l;
            const bool ok = i_live && m < p.M;
            if (GATHER == G_LINEAR) {
                a_y[j] = ok ? 0 : -100000;
                a_x[j] = 0;
                a_voff[j] = (unsigned)m * (unsigned)(p.lda * ESZ) + d_kbytes;
            } else {
                const int hw = p.H * p.Wd;
                const int b = m / hw, rr = m - b * hw;
                a_y[j] = ok ? rr / p.Wd : -100000;
                a_x[j] = rr - (rr / p.Wd) * p.Wd;
                a_voff[j] = (unsigned)m * (unsigned)(p.Cin * ESZ) + d_kbytes;  // NHWC pixel origin
            }
        }
        w_voff = (unsigned)(n0 + 8 * (wv + 8 * JA - NA) + d_l) * (unsigned)(p.ldw * ESZ) + d_kbytes;  // n < N: N % BN == 0
        i_step = 0;
        i_tap = 0;
        i_c0 = i_cbeg;
    };
    auto issue_instr = [&](int buf, int j) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + (wv + 8 * j) * 1024;
        if (j < JA) {
            int dy = 0, dx = 0;
            if (GATHER == G_CONV3) {
                dy = i_tap / 3 - 1;
                dx = i_tap - (i_tap / 3) * 3 - 1;
            } else if (GATHER == G_DECONV) {
                dy = (i_tap >> 1) - 1 + i_py;
                dx = (i_tap & 1) - 1 + i_px;
            }
            const int jj = j < JA ? j : 0;
            const int yy = a_y[jj] + dy, xx = a_x[jj] + dx;
            const bool ok = GATHER == G_LINEAR ? a_y[jj] >= 0 : (yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd);
            const int tap_off = GATHER == G_LINEAR ? i_c0 * ESZ : ((dy * p.Wd + dx) * p.Cin + i_c0) * ESZ;
            const unsigned va = ok ? (unsigned)((int)a_voff[jj] + tap_off) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, va, 0, 0, 0);
        } else {
            const unsigned kb = (unsigned)((i_tap * p.Cin + i_c0) * ESZ) + (unsigned)((j - JA) * 64) * (unsigned)(p.ldw * ESZ);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, i_live ? w_voff + kb : OOB, 0, 0, 0);
        }
    };
    const int ntaps = p.K / cslice;
    auto advance_cursor = [&]() __attribute__((always_inline)) {
        // channel block -> taps (tap_inner: the nine (four) shifted reads of a 128-byte column block of the tile's pixels follow each
        // other directly - the re-reads of the gather hit in L2 whatever else the XCD's workgroups stream meanwhile) or tap ->
        // channel blocks. Written as straight-line selects ON PURPOSE: as two branches with `i_tap = ...` / `i_c0 = ...` in each, the
        // compiler sank the stores into one store through a selected pointer, which pins the cursor in SCRATCH memory - every stage of
        // every instantiation then began with two scratch loads and an s_waitcnt vmcnt(0) in front of its DMA issue (rounds 2 - 3;
        // found in round 4 from -Rpass-analysis=kernel-resource-usage: ScratchSize 12 - 64 bytes per lane at 180 - 250 registers).
        const bool ti = p.tap_inner != 0;
        const int tap = i_tap + (ti ? 1 : 0), c0 = i_c0 + (ti ? 0 : KS);
        const bool wrap = ti ? tap == ntaps : c0 == i_cbeg + cslice;
        i_tap = ti ? (wrap ? 0 : tap) : tap + (wrap ? 1 : 0);
        i_c0 = ti ? c0 + (wrap ? KS : 0) : (wrap ? i_cbeg : c0);
        if (++i_step == nsteps) {
            i_tile += gridDim.x;
            setup_issue_tile();
        }
    };
    (void)cps;

    // ---- fragment addresses: hi halves in chunk f_kg, lo halves in chunk 4 + f_kg of the row (swizzled by row & 7)
    const int sw = f_row & 7;
    const int a_frag_off = (rg * (BM / 2) + f_row) * 128;
    const int w_frag_off = BM * 128 + (cg * (BN / 4) + f_row) * 128;
    auto frag_a = [&](int buf, int lo, int rf) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + (((lo * 4 + f_kg) ^ sw) << 4) + a_frag_off + rf * 2048);
    };
    auto frag_w = [&](int buf, int lo, int cf) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + (((lo * 4 + f_kg) ^ sw) << 4) + w_frag_off + cf * 2048);
    };

    if ((int)blockIdx.x >= ntiles) return;
    // (tried: every other workgroup of an XCD starting 8 / 16 us late so that the epilogue store bursts of the two groups
    // fall into each other's main loops - qkv 90 -> 96 / 103 us: the delay just adds, the lockstep stores are not the limit)

    // HEAD: this wave's share of the 1x1 weights (channels 64 cg .. + 63: two K = 32 blocks x two 16-map fragments x (hi, lo))
    // stays in registers for the whole launch; the 1x1 bias goes to LDS behind the partial sums (a vector load in the epilogue
    // would queue behind the next tile's stages: vmcnt retires in order). Requested before the first DMA piece.
    constexpr int HP_PITCH = 100, HP_MAXN = 28;                    // partial sums: [4 column groups][head_n][96 pixels], pitch 100
    constexpr int OFF_HEADB = 4 * HP_MAXN * HP_PITCH * 4;          // 44 800 B into the staging region
    static_assert(!HEAD || (SPLIT && GATHER == G_DECONV && RF == 6 && CF == 4 && NST == 2 && OFF_HEADB + 128 <= CST_BYTES), "fused 1x1 head");
    u32x4 hw[HEAD ? 2 : 1][2][2];
    if constexpr (HEAD) {
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    hw[kb2][nf][hl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.head_w) +
                                                                      ((size_t)(((cg * 2 + kb2) * 2 + nf) * 2 + hl) * 64 + lane) * 16);
        if (tid < HP_MAXN) reinterpret_cast<float*>(smem + OFF_CST + OFF_HEADB)[tid] = tid < p.head_n ? p.head_b[tid] : 0.f;
    }
    setup_issue_tile();
#pragma unroll
    for (int s = 0; s < NST; ++s) {
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_instr(s, j);
        advance_cursor();
    }
    wait_vm_lgkm<(NST - 1) * NI>();  // the first stage has landed
    __builtin_amdgcn_s_barrier();
    int cb = 0;  // ring buffer of the stage being consumed (runs on across tiles)
    u32x4 ah[RF], wh[CF], al[RF], wl[CF];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf) wh[cf] = frag_w(0, 0, cf);
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) ah[rf] = frag_a(0, 0, rf);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[CF][RF];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) acc[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The bias of this wave's columns is fetched HERE, not in the epilogue: a vector load there queues behind the next
        // tile's stages already in flight and behind the stores of the quarter before (vmcnt retires in order) - 5 - 9 % of
        // the f16x3 Linear launches (PSPLIT_DBG 1). The K loop's first wait covers it.
        f32x4 bias_r[CF];
        {
            int zb, mb, nb0;
            decode_tile(tile, zb, mb, nb0);
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
                bias_r[cf] = (p.bias && !(DBG & 1)) ? *reinterpret_cast<const f32x4*>(p.bias + (size_t)zb * p.strideBias_z + nb0 + cg * (BN / 4) + cf * 16 + f_kg * 4)
                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int k = 0; k < nsteps; ++k) {
            const int nb = cb + 1 == NST ? 0 : cb + 1;  // buffer of the next stage
            // ---- first half: hi x hi (bf16: the first k-block), the other half's fragments of this stage arrive underneath
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) wl[cf] = frag_w(cb, 1, cf);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) al[rf] = frag_a(cb, 1, rf);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) acc[cf][rf] = mma<SPLIT>(wh[cf], ah[rf], acc[cf][rf]);
            __builtin_amdgcn_sched_barrier(0);
            // every wave holds the rest of this stage in registers -> its buffer is free; the next stage must have landed:
            // two stages: it is the only one outstanding; three: the newest one (NI instructions per wave) may still fly
            wait_vm_lgkm<NST == 2 ? 0 : NI>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- second half: refill the freed buffer (three stages: not after a tile's last stage - the epilogue stages
            // through that buffer first); lo x hi, then hi x lo (bf16: the second k-block), the next stage's first fragments
            // replace the dying ones
            if (NST == 2 || k + 1 < nsteps) {
#pragma unroll
                for (int j = 0; j < NI; ++j) issue_instr(cb, j);
                advance_cursor();
            }
            if constexpr (SPLIT) {
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
                    for (int cf = 0; cf < CF; ++cf) acc[cf][rf] = mma<SPLIT>(wl[cf], ah[rf], acc[cf][rf]);
                    ah[rf] = frag_a(nb, 0, rf);
                }
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) acc[cf][rf] = mma<SPLIT>(wh[cf], al[rf], acc[cf][rf]);
                    wh[cf] = frag_w(nb, 0, cf);
                }
            } else {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) wh[cf] = frag_w(nb, 0, cf);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) ah[rf] = frag_a(nb, 0, rf);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int cf = 0; cf < CF; ++cf) acc[cf][rf] = mma<SPLIT>(wl[cf], al[rf], acc[cf][rf]);
            }
            cb = nb;
        }
        const int last_buf = cb == 0 ? NST - 1 : cb - 1;  // buffer of the stage consumed last: free

        if (DBG & 128) {  // dev: no epilogue (one store that never happens keeps the accumulators alive)
            f32x4 sum = acc[0][0];
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) sum += acc[cf][rf];
            if (p.M < 0) reinterpret_cast<f32x4*>(p.C)[tid] = sum;
            if (NST == 3) {
#pragma unroll
                for (int j = 0; j < NI; ++j) issue_instr(last_buf, j);
                advance_cursor();
            }
            continue;
        }
        // ---- epilogue (lane id laundered: the addresses below must not be hoisted above the K loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int e_row = lane_e & 15, e_kg = lane_e >> 4, tid_e = (tid & ~63) | lane_e;
        int z, m0, n0;
        decode_tile(tile, z, m0, n0);
        if constexpr (HEAD) {
            float* part = reinterpret_cast<float*>(smem + OFF_CST);
            const float* hb = reinterpret_cast<const float*>(smem + OFF_CST + OFF_HEADB);
            const int hwp = p.H * p.Wd;
            const int phase = p.py < 0 ? z : 2 * p.py + p.px;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (rg == r) {
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) {
                        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb2 = 0; kb2 < 2; ++kb2) {
                            f32x4 v0 = acc[2 * kb2][rf] + bias_r[2 * kb2], v1 = acc[2 * kb2 + 1][rf] + bias_r[2 * kb2 + 1];
                            f16x8 ph, pl;
                            {   // (hi, lo) of the eight ReLU'd values by split_pair: 16 VALU instructions instead of ~28
                                u32x4 phu, plu;
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    unsigned h_, l_;
                                    split_pair(relu_keep_nan(v0[2 * j]), relu_keep_nan(v0[2 * j + 1]), h_, l_);
                                    phu[j] = h_; plu[j] = l_;
                                    split_pair(relu_keep_nan(v1[2 * j]), relu_keep_nan(v1[2 * j + 1]), h_, l_);
                                    phu[2 + j] = h_; plu[2 + j] = l_;
                                }
                                ph = __builtin_bit_cast(f16x8, phu);
                                pl = __builtin_bit_cast(f16x8, plu);
                            }
                            o0 = split_mma(__builtin_bit_cast(f16x8, hw[kb2][0][0]), __builtin_bit_cast(f16x8, hw[kb2][0][1]), ph, pl, o0);
                            o1 = split_mma(__builtin_bit_cast(f16x8, hw[kb2][1][0]), __builtin_bit_cast(f16x8, hw[kb2][1][1]), ph, pl, o1);
                        }
                        const int px = rf * 16 + e_row;  // lane: pixel px of the row half, maps 4 fg + i (o0) and 16 + 4 fg + i (o1)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int n0_ = 4 * e_kg + i, n1_ = 16 + 4 * e_kg + i;
                            if (n0_ < p.head_n) part[(cg * HP_MAXN + n0_) * HP_PITCH + px] = o0[i];
                            if (n1_ < p.head_n) part[(cg * HP_MAXN + n1_) * HP_PITCH + px] = o1[i];
                        }
                    }
                }
                wait_vm_lgkm<63>();  // LDS writes only
                __builtin_amdgcn_s_barrier();
                for (int idx = tid_e; idx < p.head_n * 96; idx += THREADS) {
                    const int n = idx / 96, px = idx - n * 96;
                    const int m = m0 + r * 96 + px;
                    if (m >= p.M) continue;
                    const float sum = ((part[(0 * HP_MAXN + n) * HP_PITCH + px] + part[(1 * HP_MAXN + n) * HP_PITCH + px]) +
                                       (part[(2 * HP_MAXN + n) * HP_PITCH + px] + part[(3 * HP_MAXN + n) * HP_PITCH + px])) + hb[n];
                    const int b = m / hwp, rr = m - b * hwp;
                    p.head_out[(((size_t)b * p.head_n + n) * 4 + phase) * hwp + rr] = sum;
                }
                wait_vm_lgkm<63>();
                __builtin_amdgcn_s_barrier();  // the partial sums are rewritten by the other row half / the next tile
            }
            continue;
        }
        char* cst = NST == 2 ? smem + OFF_CST : smem + last_buf * STAGE;
        constexpr int ROWB = BN * 4;            // bytes per staged fp32 row
        constexpr int LPR = BN / 8;             // lanes per row, 8 elements each (32 or 24)
        constexpr int RPP = THREADS / LPR;      // rows per pass (16 or 21)
        constexpr int QR = BM / 4;              // rows per quarter (48 or 64)
        // bf16 Linear layers with an fp32 residual (ViT-B projection / fc2): the residual rows of a quarter are requested
        // before that quarter is staged; fetched inside the store loop, every row pass is its own load -> wait -> store
        // round trip (proj at bs 64: 167 us with, 117 us without the loads - PSPLIT_DBG 2)
        constexpr bool RES_PREFETCH = !SPLIT && GATHER == G_LINEAR;
        constexpr int NIT = (QR + RPP - 1) / RPP;
        f32x4 resv[RES_PREFETCH ? NIT : 1][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (RES_PREFETCH && p.residual && !(DBG & 2)) {
                const int cl = tid_e % LPR, rl = tid_e / LPR;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int ml = it * RPP + rl;
                    const int m = m0 + q * QR + ml;
                    if (rl < RPP && ml < QR && m < p.M) {
                        const float* r = p.residual + (p.res_mod > 0 ? (size_t)(m % p.res_mod) * p.ldres : (size_t)m * p.ldres) + n0 + cl * 8;
                        resv[it][0] = *reinterpret_cast<const f32x4*>(r);
                        resv[it][1] = *reinterpret_cast<const f32x4*>(r + 4);
                    }
                }
            }
            if (rg == (q >> 1)) {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const int nl = cg * (BN / 4) + cf * 16 + e_kg * 4;
                    const f32x4 bv = bias_r[cf];
#pragma unroll
                    for (int r2 = 0; r2 < RF / 2; ++r2) {
                        const int rf = (q & 1) * (RF / 2) + r2;
                        f32x4 v = acc[cf][rf] * p.w_inv + bv;  // (w_inv: power-of-two scale of a split-fp16 Linear layer's weights, 1 otherwise)
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = SPLIT ? relu_keep_nan(v[j]) : fmaxf(v[j], 0.f);
                        } else if (p.act == ACT_GELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = SPLIT ? gelu_erfc_as(v[j]) : gelu_erf_exact(v[j]);
                        }
                        const int ml = r2 * 16 + e_row;
                        *reinterpret_cast<f32x4*>(cst + ml * ROWB + ((((nl >> 2)) ^ (ml & 7)) << 4)) = v;
                    }
                }
            }
            wait_vm_lgkm<63>();  // LDS writes only: the DMA of the next tile stays in flight
            __builtin_amdgcn_s_barrier();
            const int cl = tid_e % LPR, rl = tid_e / LPR;
            if constexpr (POOL) {
                static_assert(!POOL || (SPLIT && GATHER == G_CONV3 && RF == 6 && NST == 3), "fused pooling: split-fp16 3x3 conv, 192-row tiles");
                // quarter q = image rows pool_h q .. (QR = pool_h * W pixels); thread (window j of the row, 8 channels): max over the
                // pool_h x pool_w staged pixels, then ReLU (bias is already in: max and + commute)
                const int Wo = p.Wd / p.pool_w;
                if (rl < Wo) {
                    f32x4 v0 = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, v1 = v0;
                    for (int r = 0; r < p.pool_h; ++r)
                        for (int c = 0; c < p.pool_w; ++c) {
                            const int ml = r * p.Wd + rl * p.pool_w + c;
                            const f32x4 a0 = *reinterpret_cast<const f32x4*>(cst + ml * ROWB + (((2 * cl) ^ (ml & 7)) << 4));
                            const f32x4 a1 = *reinterpret_cast<const f32x4*>(cst + ml * ROWB + (((2 * cl + 1) ^ (ml & 7)) << 4));
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                v0[i] = fmaxf(v0[i], a0[i]);
                                v1[i] = fmaxf(v1[i], a1[i]);
                            }
                        }
                    f16x8 hv, lv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float b0 = fmaxf(v0[i], 0.f), b1 = fmaxf(v1[i], 0.f);
                        hv[i] = split_hi(b0);
                        lv[i] = split_lo(b0, hv[i]);
                        hv[4 + i] = split_hi(b1);
                        lv[4 + i] = split_lo(b1, hv[4 + i]);
                    }
                    const int b_img = m0 / (p.H * p.Wd);
                    const size_t eoff = (size_t)z * p.strideC_z + (((size_t)b_img * 4 + q) * Wo + rl) * p.ldc + n0 + cl * 8;
                    if (m0 < p.M) {
                        char* o = split_addr(p.C, eoff);
                        *reinterpret_cast<f16x8*>(o) = hv;
                        *reinterpret_cast<f16x8*>(o + 64) = lv;
                    }
                }
            } else if (rl < RPP) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int ml = it * RPP + rl;
                    const int m = m0 + q * QR + ml;
                    if (ml >= QR || m >= p.M) continue;
                    size_t orow = m;
                    if (GATHER == G_DECONV) {  // phase-interleaved output pixel (2y+py, 2x+px) of a (2H, 2W) map
                        const int hw = p.H * p.Wd;
                        const int b = m / hw, r = m - b * hw;
                        const int y = r / p.Wd, x = r - y * p.Wd;
                        const int py = p.py < 0 ? (z >> 1) : p.py, px = p.py < 0 ? (z & 1) : p.px;
                        orow = ((size_t)b * (2 * p.H) + 2 * y + py) * (2 * p.Wd) + 2 * x + px;
                    }
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(cst + ml * ROWB + (((2 * cl) ^ (ml & 7)) << 4));
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(cst + ml * ROWB + (((2 * cl + 1) ^ (ml & 7)) << 4));
                    const size_t eoff = (size_t)z * p.strideC_z + orow * p.ldc + n0 + cl * 8;
                    if (RES_PREFETCH) {
                        if (p.residual && !(DBG & 2)) {
                            v0 += resv[it][0];
                            v1 += resv[it][1];
                        }
                    } else if (p.residual && !(DBG & 2)) {  // fp32, same indexing as the output, or a (res_mod, N) table broadcast over the rows
                        const float* r = p.residual + (p.res_mod > 0 ? (size_t)(m % p.res_mod) * p.ldres : orow * p.ldres) + n0 + cl * 8;
                        v0 += *reinterpret_cast<const f32x4*>(r);
                        v1 += *reinterpret_cast<const f32x4*>(r + 4);
                    }
                    if (p.out_bf16 == 1) {
                        const bf16x8 ov = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3],
                                           (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.C) + eoff) = ov;
                    } else if (p.out_bf16 == 2) {
                        f16x8 hv, lv;
                        {
                            u32x4 hu, lu;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                unsigned h_, l_;
                                split_pair(v0[2 * j], v0[2 * j + 1], h_, l_);
                                hu[j] = h_; lu[j] = l_;
                                split_pair(v1[2 * j], v1[2 * j + 1], h_, l_);
                                hu[2 + j] = h_; lu[2 + j] = l_;
                            }
                            hv = __builtin_bit_cast(f16x8, hu);
                            lv = __builtin_bit_cast(f16x8, lu);
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
            }
            wait_vm_lgkm<63>();
            __builtin_amdgcn_s_barrier();  // the staging region is reused by the next quarter / the next tile
        }
        // stores and loads share vmcnt but may retire out of order with respect to each other: drain before the counted
        // waits of the next tile rely on the count again (three-stage form; the two-stage K loop waits vmcnt(0) itself, there
        // the stores retire under the next tile's first step)
        if (NST == 3) wait_vm_lgkm<0>();
        if (NST == 3) {  // the refill deferred above (every wave is past the last barrier of the staging region)
#pragma unroll
            for (int j = 0; j < NI; ++j) issue_instr(last_buf, j);
            advance_cursor();
        }
    }
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

}  // namespace psplit

// which tile shape serves this problem, or 0: 1 = 192 x 256 (deconvolutions, N % 256), 2 = 256 x 192 (N % 192);
// three-stage forms (PP_PSPLIT_NST=3, dev): 3 = 128 x 256, 4 = 192 x 192
static int psplit_nst() {  // 0 (default): by shape, see panel_split_shape; 2 / 3 force the two- / three-stage forms (dev)
    const int v = option("psplit_nst");
    return v;
}
static int panel_split_shape(const GemmParams& p, int groups) {
    const bool three = psplit_nst() == 3;
    if (p.gather == G_DECONV) return p.N % 256 == 0 ? ((three && !p.head_w) ? 3 : 1) : 0;
    if (p.N % 192 != 0) return 0;
    if (three || p.pool_h > 0) return 4;
    // Linear layers: the persistent grid walks whole rounds of tiles (one per CU); the 192 x 192 three-stage form wins where
    // it turns a ragged last round into full ones (qkv at bs 64: 576 tiles of 256 x 192 = 2.25 rounds -> 768 tiles of
    // 192 x 192 = 3.0: 91 -> 79 us) and loses otherwise (fc1: 3 rounds either way but 12 % more fill per FLOP: 117 -> 125 us)
    if (p.gather == G_LINEAR && psplit_nst() != 2) {
        const long long cus = 256;
        const long long t2 = (long long)(p.N / 192) * ((p.M + 255) / 256) * groups, t4 = (long long)(p.N / 192) * ((p.M + 191) / 192) * groups;
        const long long c2 = ((t2 + cus - 1) / cus) * 256 * 192, c4 = ((t4 + cus - 1) / cus) * 192 * 192;
        if (c4 * 10 < c2 * 9) return 4;
    }
    return 2;
}

// bf16 Linear layers shorter than this stay on the 128 x 128 kernel, whose co-resident workgroups overlap one tile's epilogue
// with another's main loop (dev: PP_PANEL_LINEAR_MINK). Measured on ViT-B at bs 64 (scripts/bench_base.py): K = 768 with bf16
// output (qkv, fc1) 8.66 ms on the 128 x 128 kernel vs 9.83 ms here; K = 768 with fp32 output + residual (proj) 6.65 vs 6.28 ms
static int panel_linear_min_k(bool out_bf16) {
    const int v = option("panel_linear_mink");
    return v > 0 ? v : (out_bf16 ? 1536 : 768);
}
static bool panel_bf16_conv() {  // dev: bf16 convolutions through this kernel instead of pp_panel_gemm.hip
    const bool v = option("psplit_bf16_conv") != 0;
    return v;
}

static void shape_dims(int shape, int& BM, int& BN) {
    BM = shape == 1 ? 192 : shape == 2 ? 256 : shape == 3 ? 128 : 192;
    BN = shape == 1 ? 256 : shape == 2 ? 192 : shape == 3 ? 256 : 192;
}

bool panel_split_supported(const GemmParams& p, int prec, int groups) {
    if (p.planar_P > 0) return false;
    if (p.ksplit > 1) {  // split-K: channel-range slices of a split-fp16 3x3 convolution walked taps-inner, fp32 partial sums out
        if (prec != PP_PREC_F16X3 || p.gather != G_CONV3 || p.out_bf16 != 0 || p.bias || p.residual || p.head_w || p.pool_h > 0 || p.act != ACT_NONE) return false;
        if (groups % p.ksplit != 0 || p.Cin % (32 * p.ksplit) != 0 || p.K != 9 * (p.Cin / p.ksplit) || option("psplit_tap_inner") < 1) return false;
    }
    if (p.pool_h > 0) {  // fused MaxPool + ReLU: one 16 x 12 image per 192-row tile, a quarter of the tile = one row of windows
        if (prec != PP_PREC_F16X3 || p.gather != G_CONV3 || p.out_bf16 != 2 || p.residual || p.head_w) return false;
        if (p.H * p.Wd != 192 || p.pool_h * p.Wd != 48 || p.H != 4 * p.pool_h || p.Wd % p.pool_w != 0 || p.M % 192 != 0 || p.N % 192 != 0) return false;
        if (p.act != ACT_NONE) return false;
    }
    if (prec == PP_PREC_F16X3) {
        if (p.head_w && !(p.gather == G_DECONV && p.N == 256 && p.head_n >= 1 && p.head_n <= 28)) return false;
        if (p.out_bf16 != 0 && p.out_bf16 != 2) return false;
        if (p.K % 32 != 0 || p.Cin % 32 != 0 || p.ldc % 32 != 0 || p.lda % 32 != 0 || p.ldw % 32 != 0) return false;
        if (p.strideA_z % 32 != 0 || p.strideW_z % 32 != 0 || p.strideC_z % 32 != 0) return false;
    } else if (prec == PP_PREC_BF16) {  // convolutions have their own kernel (pp_panel_gemm.hip: fused head, split-K partials)
        if (p.gather == G_LINEAR ? (p.K < panel_linear_min_k(p.out_bf16 != 0)) : !panel_bf16_conv()) return false;
        if (p.K % 128 != 0 || p.Cin % 64 != 0 || p.head_w) return false;
        if (p.out_bf16 != 0 && p.out_bf16 != 1) return false;
        if (p.ldc % 8 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0) return false;
    } else {
        return false;
    }
    if (p.residual && (p.out_bf16 == 2 || (p.ldres % 4) != 0)) return false;
    const int shape = panel_split_shape(p, groups);
    if (!shape) return false;
    int BM, BN;
    shape_dims(shape, BM, BN);
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    return ntiles >= 192;  // one workgroup per CU: with fewer tiles the 128 x 128 kernel spreads the work better
}

int panel_split_gemm(const GemmParams& p_in, int prec, int groups, hipStream_t s) {
    using namespace psplit;
    GemmParams p = p_in;
    p.groups = groups;
    // 3x3 convolutions: taps inner + weight-set-major tiles (first tower stage at bs 64: 765 -> 371 MB fetched per launch, 635 ->
    // 630 us). Weight-major WITHOUT taps inner is the worst of the four (2.46 GB: 32 row panels live per XCD, the nine shifted
    // re-reads of each miss L2), so it is only honoured together. Deconvolutions measured 1 - 3 % slower taps-inner: off.
    const int ti = option("psplit_tap_inner");
    p.tap_inner = (p.gather == G_CONV3 && ti >= 1) || (p.gather == G_DECONV && ti >= 2) ? 1 : 0;
    PP_REQUIRE(p.ksplit <= 1 || p.tap_inner, PP_ERR_UNSUPPORTED, "pp panel split gemm: split-K slices are channel ranges walked taps-inner");
    p.tile_order = (p.gather == G_CONV3 && p.tap_inner && option("psplit_conv_weight_major") != 0) ? 1 : 0;
    if (p.gather == G_DECONV && option("psplit_deconv_weight_major") != 0) p.tile_order = 1;  // dev A/B (is deconv2 + 1x1 traffic-bound?)
    if (p.gather == G_LINEAR) p.Cin = p.K;
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < OOB && p.w_bytes < OOB, PP_ERR_UNSUPPORTED,
               "pp panel split gemm: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    void (*kern)(const GemmParams) = nullptr;
    const int shape = panel_split_shape(p, groups);
    int BM, BN;
    shape_dims(shape, BM, BN);
    const bool sp = prec != PP_PREC_BF16;
    if ((long long)p.K * BN * (sp ? 4 : 2) > 3 * 1024 * 1024) p.tile_order = 0;  // a weight set must stay in the XCD's 4 MB L2 beside the streamed rows (ViT-B towers: 5.3 MB)
#define PP_PS(G, RF, CF, NST) (sp ? panel_split_kernel<G, RF, CF, true, NST> : panel_split_kernel<G, RF, CF, false, NST>)
    switch (p.gather) {
        case G_DECONV:
            if (p.head_w) {
                PP_REQUIRE(sp && shape == 1, PP_ERR_UNSUPPORTED, "pp panel split gemm: the fused 1x1 head needs the split-fp16 192 x 256 form");
                kern = panel_split_kernel<G_DECONV, 6, 4, true, 2, true>;
            } else {
                kern = shape == 3 ? PP_PS(G_DECONV, 4, 4, 3) : PP_PS(G_DECONV, 6, 4, 2);
            }
            break;
        case G_CONV3:
            if (p.pool_h > 0) {
                PP_REQUIRE(sp && shape == 4, PP_ERR_UNSUPPORTED, "pp panel split gemm: fused pooling needs the split-fp16 192 x 192 form");
                kern = panel_split_kernel<G_CONV3, 6, 3, true, 3, false, true>;
            } else {
                kern = shape == 4 ? PP_PS(G_CONV3, 6, 3, 3) : PP_PS(G_CONV3, 8, 3, 2);
            }
            break;
        default: kern = shape == 4 ? PP_PS(G_LINEAR, 6, 3, 3) : PP_PS(G_LINEAR, 8, 3, 2); break;
    }
#undef PP_PS
    int slots = device_cus();
    slots -= slots % 8;
    // Ragged last round of a Linear layer (one persistent workgroup per CU): the ViT-B projection / fc2 at bs 64 are 864 tiles of
    // 256 x 192 = 3.375 rounds - the fourth round keeps 96 of 256 CUs busy for a whole tile time. The rows of the whole rounds go to this
    // kernel, the tail rows to a second launch on 128 x 192 tiles (half the tile time, twice the tiles): 3 + ~0.5 rounds instead of 4
    // (no 8-wave tile shape makes whole rounds at M = 2^11 x 27). Only where the tail fits ONE round of the small tiles.
    if (p.gather == G_LINEAR && sp && shape == 2 && groups == 1 && p.res_mod == 0 && option("psplit_tail") != 0) {
        const long long ntn = p.N / BN, ntm = (p.M + BM - 1) / BM, tiles = ntn * ntm;
        const long long rounds = tiles / slots, rem = tiles % slots;
        const long long m1 = (rounds * slots / ntn) * BM;  // rows of the whole rounds
        const long long tail_tiles = m1 < p.M ? ntn * ((p.M - m1 + 127) / 128) : 0;
        if (rounds >= 1 && rem > 0 && rem * 4 <= (long long)slots * 3 && tail_tiles > 0 && tail_tiles <= slots && m1 > 0) {
            GemmParams p1 = p, p2 = p;
            p1.M = (int)m1;
            const size_t a_off = (size_t)m1 * p.lda * 4;
            p2.M = p.M - (int)m1;
            p2.A = reinterpret_cast<const char*>(p.A) + a_off;
            p2.a_bytes = p.a_bytes - (unsigned)a_off;
            p2.C = reinterpret_cast<char*>(p.C) + (size_t)m1 * p.ldc * 4;  // fp32 and split rows are both 4 bytes per element
            if (p.residual) p2.residual = p.residual + (size_t)m1 * (p.ldres ? p.ldres : p.ldc);
            void (*k1)(const GemmParams) = panel_split_kernel<G_LINEAR, 8, 3, true, 2>;
            void (*k2)(const GemmParams) = panel_split_kernel<G_LINEAR, 4, 3, true, 2>;
            const int lds1 = lds_bytes(256, 192, 2), lds2 = lds_bytes(128, 192, 2);
            PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
            PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
            hipLaunchKernelGGL(k1, dim3(slots), dim3(THREADS), lds1, s, p1);
            PP_LAUNCH_CHECK();
            hipLaunchKernelGGL(k2, dim3((unsigned)tail_tiles), dim3(THREADS), lds2, s, p2);
            PP_LAUNCH_CHECK();
            return PP_OK;
        }
    }
    const int lds = lds_bytes(BM, BN, shape >= 3 ? 3 : 2);
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    PP_REQUIRE(ntiles < (1ll << 30), PP_ERR_UNSUPPORTED, "pp panel split gemm: too many output tiles");
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp
