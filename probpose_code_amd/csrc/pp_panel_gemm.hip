// Wide-tile implicit-GEMM convolutions for gfx950 (bf16 operands, bf16 output): the two deconvolutions of the heatmap
// branch (ConvTranspose2d k4 s2 p1 + BN + ReLU, probmap_head.py:435-472) and the 3x3 tower convolutions
// (probmap_head.py:261-294) - 570 of the path's 1720 GFLOP at bs 64.
//
// The 128 x 128 tiles of pp_gemm.hip move 32 KiB from L2 into LDS per 2.1 MFLOP; at K = 1024 ... 3456 that kernel runs
// at a quarter of the MFMA rate and the L2 -> LDS fill is what it waits for. This kernel owns a whole CU per workgroup
// and uses 192 x 256 (deconv) / 256 x 192 (conv 3x3) tiles: 56 KiB per 6.3 MFLOP, 1.7x fewer fill bytes per FLOP.
//
//   * 512 threads = 8 waves = two per SIMD, wave (rg, cg): row half rg, column quarter cg; wave tile RF x CF MFMA
//     16x16x32 fragments (6 x 4 or 8 x 3 = 24 MFMAs per 32 elements of K), fp32 accumulators (96 VGPRs);
//   * a stage is 64 elements of K: activation tile BM rows x 128 B + weight tile BN rows x 128 B = 56 KiB, the same
//     XOR-swizzled LDS image as everywhere else, filled by LDS-DMA. Two stages. Rows are fetched as whole 128-byte
//     segments: with 64-byte segments the DMA stream tops out at 13 TB/s instead of 21 (scripts/micro/dma_pattern.hip);
//   * fragments are double-buffered in registers at half-stage granularity: while the MFMAs of (stage s, first half)
//     run, the second half is read; ONE barrier per stage sits between the halves - there every wave holds the rest of
//     stage s in registers, so its buffer takes the DMA of stage s+2 at once, and stage s+1 has landed (vmcnt 0).
//     The stream of stages runs on across output tiles (persistent workgroups, one per CU): the next tile's first
//     stages land under the current tile's last MFMAs and its epilogue;
//   * inside a half the LDS reads of the next fragments and the DMA instructions are interleaved with the MFMAs
//     (sched_group_barrier), see pp_mlp.hip for the measurements behind that;
//   * implicit im2col: a DMA lane computes its pixel / tap address itself, out-of-image taps and tail rows use an
//     out-of-bounds buffer offset (the DMA writes zeros);
//   * epilogue: + bias (folded BN), optional ReLU, bf16, staged through a separate 48 KiB LDS region one row half at a
//     time and written as contiguous 16-byte lane stores (deconv: to the phase-interleaved output pixel).
#include "pp_common.h"
#include "pp_gemm.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace panel {

#ifndef PANEL_DBG
#define PANEL_DBG 0
#endif
// dev ablation switches (scripts/micro/panel_ablate.sh), 0 in the product build: 1 DMA out of bounds (no traffic),
// 4 no MFMA, 8 no DMA, 16 no LDS fragment reads, 32 activation DMA out of bounds only, 64 weight DMA out of bounds only,
// 128 no epilogue, 256 clock probe, 512 fused head without its weight reads
constexpr int DBG = PANEL_DBG;

constexpr int THREADS = 512;
constexpr int STAGE = 56 * 1024, NSTAGE = 2;
constexpr int OFF_CST = NSTAGE * STAGE;  // 112 KiB
constexpr int CST_BYTES = 48 * 1024;
constexpr int LDS = OFF_CST + CST_BYTES;  // 160 KiB
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS == 160 * 1024, "LDS map");

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x008, SG_VMEM = 0x010, SG_DS_READ = 0x100;

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}

// s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier as builtins (the compiler's wait-count bookkeeping sees them); gfx9
// encoding: vmcnt [3:0] + [15:14], expcnt [6:4] = 7 (none), lgkmcnt [11:8]
template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

template <int GATHER, int RF, int CF>
__global__ __launch_bounds__(THREADS, 2) void panel_gemm_kernel(const GemmParams p) {
    constexpr int BM = 32 * RF, BN = 64 * CF;
    constexpr int NA = BM / 8;            // DMA instructions of the activation tile; the weight tile takes the other 56 - NA
    constexpr int JA = NA / 8;            // of a wave's seven instructions, j < JA fetch activation rows
    static_assert(BM * 128 + BN * 128 == STAGE, "a stage is 56 KiB");
    static_assert((BM / 2) * BN * 2 == CST_BYTES, "one row half of the bf16 output tile fills the staging region");
    static_assert(NA % 8 == 0 && JA <= 4, "instruction split");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;

    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntn * ntm * p.groups;
    const int nsteps = p.K / 64;  // K-steps per output tile (even)

    // Tile order. Block b runs on XCD b % 8 (private 4 MiB L2) and visits b, b + G, ... (G % 8 == 0 keeps it there);
    // logical tile = (t % 8) * (ntiles / 8) + t / 8 gives each XCD a contiguous run of the tile list, and the list is
    // ordered row panel -> group -> column tile: the groups of these launches (four towers / four deconv phases) read
    // the SAME activations, so the 32 workgroups of an XCD share 4 - 8 activation panels, and because they sweep K in
    // step, only the weight slices of the current taps are hot in that L2 at any moment.
    auto decode_tile = [&](int t, int& z, int& m0, int& n0) {
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        n0 = (t % ntn) * BN;
        const int r = t / ntn;
        z = r % p.groups;
        m0 = (r / p.groups) * BM;
    };

    // ---- the DMA side: a cursor (tile, K-step) that runs two stages ahead of the MFMAs.
    // Instruction i of a stage moves LDS lines 8 i .. 8 i + 7 (lines 0 .. BM-1: activation rows, then weight rows).
    // Wave w issues i = w + 8 j, j = 0..6; lane (line l, physical chunk pc) lands at byte 16 pc of its line and
    // therefore fetches logical chunk pc ^ (l & 7). Every row contributes one whole 128-byte segment per stage:
    // with 64-byte segments (K-steps of 32) the same DMA stream moves 13 TB/s instead of 21 (scripts/micro/dma_pattern.hip).
    const int d_l = lane >> 3;
    const unsigned d_kbytes = (unsigned)(((lane & 7) ^ d_l) << 4);
    unsigned a_voff[JA];
    int a_yx[JA];     // pixel of the lane's activation row: y << 16 | x, y = -30000 for tail rows (fails every bounds test)
    unsigned w_voff;  // weight row of instruction j = JA; the later ones are 64 rows further each
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    int i_tile = blockIdx.x, i_step = 0, i_tap = 0, i_c0 = 0, i_py = 0, i_px = 0, i_tap_lo = 0;
    bool i_live = true;
    // (A per-workgroup rotation of the K sweep - the layer kernel's 7 % - was measured 2 - 12 % SLOWER here and is gone: these workgroups of an XCD
    //  do not run in step.)
    auto setup_issue_tile = [&]() {
        int z = 0, m0 = 0, n0 = 0;
        i_live = i_tile < ntiles;
        if (i_live) decode_tile(i_tile, z, m0, n0);
        int tap0 = 0;
        if (GATHER == G_CONV3 && p.ksplit > 1) {  // split-K over whole taps: group index = slice * problems + problem
            const int problems = p.groups / p.ksplit;
            tap0 = (z / problems) * (9 / p.ksplit);
            z = z % problems;
        }
        if (GATHER == G_DECONV) {
            i_py = p.py < 0 ? (z >> 1) : p.py;
            i_px = p.py < 0 ? (z & 1) : p.px;
        }
        const char* Act = reinterpret_cast<const char*>(p.A) + (size_t)z * p.strideA_z * 2;
        const char* Wt = reinterpret_cast<const char*>(p.W) + (size_t)z * p.strideW_z * 2;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Act), 0, p.a_bytes, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, p.w_bytes, 0x00020000);
        const int hw = p.H * p.Wd;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int m = m0 + 8 * (wv + 8 * j) + d_l;
            const int b = m / hw, rr = m - b * hw;
            const int y = (i_live && m < p.M) ? rr / p.Wd : -30000;
            const int x = rr - (rr / p.Wd) * p.Wd;
            a_yx[j] = (y << 16) | x;
            a_voff[j] = (unsigned)m * (unsigned)(p.Cin * 2) + d_kbytes;  // NHWC pixel origin
        }
        w_voff = (unsigned)(n0 + 8 * (wv + 8 * JA - NA) + d_l) * (unsigned)(p.ldw * 2) + d_kbytes;  // n < N: N % BN == 0
        i_step = 0;
        i_tap_lo = tap0;
        i_tap = tap0;
        i_c0 = 0;
    };
    // issue instruction j of the stage at the cursor into ring buffer `buf`
    auto issue_instr = [&](int buf, int j) {
        if (DBG & 8) return;
        char* dst = smem + buf * STAGE + (wv + 8 * j) * 1024;
        if (j < JA) {
            int dy, dx;
            if (GATHER == G_CONV3) {  // 3x3, pad 1: tap = ky*3 + kx reads (y + ky - 1, x + kx - 1)
                dy = i_tap / 3 - 1;
                dx = i_tap - (i_tap / 3) * 3 - 1;
            } else {  // deconv k4 s2 p1, output phase (py, px): tap = ty*2 + tx reads (y + ty - 1 + py, x + tx - 1 + px)
                dy = (i_tap >> 1) - 1 + i_py;
                dx = (i_tap & 1) - 1 + i_px;
            }
            const int yy = (a_yx[j < JA ? j : 0] >> 16) + dy, xx = (a_yx[j < JA ? j : 0] & 0xffff) + dx;
            const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
            const int tap_off = ((dy * p.Wd + dx) * p.Cin + i_c0) * 2;
            const unsigned va = (ok && !(DBG & (1 | 32))) ? (unsigned)((int)a_voff[j < JA ? j : 0] + tap_off) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, va, 0, 0, 0);
        } else {
            const unsigned kb = (unsigned)((i_tap * p.Cin + i_c0) * 2) + (unsigned)((j - JA) * 64) * (unsigned)(p.ldw * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, (i_live && !(DBG & (1 | 64))) ? w_voff + kb : OOB, 0, 0, 0);
        }
    };
    auto advance_cursor = [&]() {
        i_c0 += 64;
        if (i_c0 == p.Cin) {
            i_c0 = 0;
            ++i_tap;
            if (i_tap == i_tap_lo + nsteps / (p.Cin / 64)) i_tap = i_tap_lo;  // wrap (rotated sweep)
        }
        if (++i_step == nsteps) {
            i_tile += gridDim.x;
            setup_issue_tile();
        }
    };

    // ---- fragment reads: K half h (32 elements) of the stage in buffer buf
    const int sw = f_row & 7;
    const int a_frag_off = (rg * (BM / 2) + f_row) * 128;
    const int w_frag_off = BM * 128 + (cg * (BN / 4) + f_row) * 128;
    auto read_frags = [&](int buf, int h, u32x4 (&af)[RF], u32x4 (&wf)[CF]) {
        const char* base = smem + buf * STAGE + (((h * 4 + f_kg) ^ sw) << 4);
        if (DBG & 16) {
            for (int cf = 0; cf < CF; ++cf) asm volatile("" : "=v"(wf[cf]));
            for (int rf = 0; rf < RF; ++rf) asm volatile("" : "=v"(af[rf]));
            return;
        }
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) wf[cf] = *reinterpret_cast<const u32x4*>(base + w_frag_off + cf * 2048);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) af[rf] = *reinterpret_cast<const u32x4*>(base + a_frag_off + rf * 2048);
    };

    if ((int)blockIdx.x >= ntiles) return;
    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if (DBG & 256) {  // dev: shader-clock cycles against the 100 MHz reference over the launch -> the clock the kernel ran at
        dbg_t0 = __builtin_readcyclecounter();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }

    // fused 1x1 convolution (pp_deconv_head): its weights (32 rows x BN bf16 = 16 KiB) stay in the last third of the staging
    // region for the whole launch, in the swizzled row image the MFMA fragments are read from
    constexpr int HEAD_ROWS = 64, OFF_HEADW = HEAD_ROWS * BN * 2;  // the head epilogue stages 64 rows at a time
    // Rows 29 - 31 of that image (pp_deconv_head takes at most 28 output maps: zero weights whose products are never stored)
    // carry the 1x1 bias and the deconvolution's folded-BN bias: read from global memory in the epilogue - or through a
    // pointer the compiler turns into a FLAT load, which counts in vmcnt too - either of them queues behind the next
    // tile's stages and the previous pass's logit stores (PANEL_DBG 1024 on the global-bias form: 32 us per launch).
    constexpr int HEAD_BIAS_ROW = 30, HEAD_B_ROW = 29;
    if (GATHER == G_DECONV && p.head_w) {
        for (int i = tid; i < HEAD_B_ROW * (BN / 8); i += THREADS) {
            const int n = i / (BN / 8), c = i - n * (BN / 8);
            const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.head_w) + (size_t)n * (BN * 2) + c * 16);
            *reinterpret_cast<u32x4*>(smem + OFF_CST + OFF_HEADW + n * (BN * 2) + ((c ^ (n & 7)) << 4)) = v;
        }
        float* brow = reinterpret_cast<float*>(smem + OFF_CST + OFF_HEADW + HEAD_BIAS_ROW * (BN * 2));
        float* hrow = reinterpret_cast<float*>(smem + OFF_CST + OFF_HEADW + HEAD_B_ROW * (BN * 2));
        for (int i = tid; i < BN; i += THREADS) brow[i] = p.bias ? p.bias[i] : 0.f;  // (one bias for all four phases, N == BN)
        for (int i = tid; i < BN / 2; i += THREADS) hrow[i] = i < p.head_n ? p.head_b[i] : 0.f;
    }

    // ---- prologue: both stages in flight, the first half of the first one into registers
    setup_issue_tile();
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
#pragma unroll
        for (int j = 0; j < 7; ++j) issue_instr(s, j);
        advance_cursor();
    }
    wait_vm_lgkm<7>();
    __builtin_amdgcn_s_barrier();
    u32x4 af[2][RF], wf[2][CF];
    read_frags(0, 0, af[0], wf[0]);

    auto mfmas = [&](f32x4 (&acc)[CF][RF], const u32x4 (&a)[RF], const u32x4 (&w)[CF]) {
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)  // row-fragment outer: an activation fragment dies after CF MFMAs
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                if (!(DBG & 4)) acc[cf][rf] = mma(w[cf], a[rf], acc[cf][rf]);
            }
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[CF][RF];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) acc[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int k2 = 0; k2 < nsteps; k2 += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // K-step k2 + s lives in ring buffer s
                // ---- first half: fragments (s, h0) are in registers; read (s, h1) under their MFMAs.
                // Issue order: the first activation fragments are dead after a few MFMAs - only then do the reads start
                // (register budget: 2 waves / SIMD = 256 VGPRs), one per two MFMAs.
                __builtin_amdgcn_sched_barrier(0);
                read_frags(s, 1, af[1], wf[1]);
                mfmas(acc, af[0], wf[0]);
                SGB(SG_MFMA, 24 - 2 * (RF + CF) + 2);
#pragma unroll
                for (int i = 0; i < RF + CF; ++i) {
                    SGB(SG_DS_READ, 1);
                    if (i < RF + CF - 1) SGB(SG_MFMA, 2);
                }
                // ---- every wave holds the second half in registers -> buffer s is free; the other buffer has landed
                __builtin_amdgcn_sched_barrier(0);
                wait_vm_lgkm<0>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- second half: refill buffer s with the stage two steps ahead, read (s + 1, h0)
#pragma unroll
                for (int j = 0; j < 7; ++j) issue_instr(s, j);
                advance_cursor();
                read_frags(s ^ 1, 0, af[0], wf[0]);
                mfmas(acc, af[1], wf[1]);
                SGB(SG_MFMA, 24 - 2 * (RF + CF) + 2);
                SGB(SG_VMEM, 1);
#pragma unroll
                for (int i = 0; i < RF + CF; ++i) {
                    SGB(SG_DS_READ, 1);
                    if (i < RF + CF - 1) SGB(SG_MFMA, 2);
                    if (i < 6) SGB(SG_VMEM, 1);
                }
            }
        }

        if (DBG & 128) {  // dev: no epilogue at all (keeps the accumulators alive through one store that never happens)
            f32x4 sum = acc[0][0];
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) sum += acc[cf][rf];
            if (p.M < 0) reinterpret_cast<f32x4*>(p.C)[tid] = sum;
            continue;
        }
        // ---- epilogue. Accumulator layout: lane holds n = 4 f_kg + (0..3) of fragment column cf for row f_row of
        // fragment row rf. One row half (one rg) at a time through the staging region, 16-byte chunks XOR-swizzled by
        // (row & 7); then every thread stores 16-byte pieces of whole output rows.
        // (lane-dependent addresses from a laundered lane id: they are loop-invariant, hoisted above the K loop they would
        // cost the registers that loop does not have)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int e_row = lane_e & 15, e_kg = lane_e >> 4, tid_e = (tid & ~63) | lane_e;
        int z, m0, n0;
        decode_tile(tile, z, m0, n0);
        if (GATHER == G_CONV3 && p.ksplit > 1) {
            // split-K partial sums: fp32, no bias / activation, straight from the accumulators (16 bytes per lane, the four
            // lane groups of a row fragment make 64 contiguous bytes); slice-major [group index][M][N]
            float* __restrict__ Cp = reinterpret_cast<float*>(p.C) + (size_t)z * p.strideC_z;
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) {
                const int m = m0 + rg * (BM / 2) + rf * 16 + e_row;
                if (m >= p.M) continue;
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
                    *reinterpret_cast<f32x4*>(Cp + (size_t)m * p.ldc + n0 + cg * (BN / 4) + cf * 16 + e_kg * 4) = acc[cf][rf];
            }
            if (!(DBG & 2048)) wait_vm_lgkm<0>();  // (stores share vmcnt with the DMA of the next tile: drain, as below)
            continue;
        }
        const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias_z : nullptr;
        __bf16* __restrict__ Cb = reinterpret_cast<__bf16*>(p.C) + (size_t)z * p.strideC_z;
        char* cst = smem + OFF_CST;
        constexpr int ROWB = BN * 2;         // bytes per staged row
        constexpr int LPR = ROWB / 16;       // 16-byte lanes per row (32 or 24)
        constexpr int RPP = THREADS / LPR;   // rows per pass (16 or 21)
        if (GATHER == G_DECONV && p.head_w) {
            // Fused 1x1 convolution (the heatmap head's final layer, probmap_head.py:244-249): a staged row holds ALL BN = N
            // channels of a pixel, so logits[pixel, n] = sum_c tile[pixel, c] Wf[n, c] + bf[n] is 2 x 8 more MFMAs per row
            // fragment and the 201 MB feature map never reaches HBM. Three passes of 64 rows (four row fragments, waves 0-3
            // one each): 32 KiB of staging + the 16 KiB of Wf resident beside it - read from L2 per tile instead they cost
            // 33 us per launch, every read queueing behind the next tile's stages (in-order vmcnt). Activations are the
            // MFMA "A" operand here: a lane ends up with 4 consecutive pixels of one channel, which is contiguous in the
            // phase-separated logits layout (B, n, phase, y * W + x).
            static_assert(GATHER != G_DECONV || (RF == 6 && BM == 192 && OFF_HEADW + 32 * BN * 2 <= CST_BYTES), "head epilogue geometry");
            const char* wfl = cst + OFF_HEADW;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    const int g = rg * RF + rf;  // row fragment of the tile (wave-uniform)
                    if ((g >> 2) != q) continue;
#pragma unroll
                    for (int cf = 0; cf < CF; ++cf) {
                        const int nl = cg * (BN / 4) + cf * 16 + e_kg * 4;
                        f32x4 v = acc[cf][rf];
                        v += *reinterpret_cast<const f32x4*>(wfl + HEAD_BIAS_ROW * ROWB + nl * 4);
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        const int ml = (g & 3) * 16 + e_row;
                        const int byte = nl * 2;
                        *reinterpret_cast<bf16x4*>(cst + ml * ROWB + ((((byte >> 4) ^ (ml & 7)) << 4) | (byte & 15))) = ov;
                    }
                }
                wait_vm_lgkm<63>();  // LDS writes only: the DMA of the next tile stays in flight
                __builtin_amdgcn_s_barrier();
                if (wv < 4) {
                    const int ml = wv * 16 + e_row;
                    f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ks = 0; ks < BN / 32; ++ks) {
                        const u32x4 a = *reinterpret_cast<const u32x4*>(cst + ml * ROWB + (((4 * ks + e_kg) ^ (ml & 7)) << 4));
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            const u32x4 b = (DBG & 512) ? u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, (unsigned)e_row}
                                                        : *reinterpret_cast<const u32x4*>(wfl + (nf * 16 + e_row) * ROWB + (((4 * ks + e_kg) ^ (e_row & 7)) << 4));
                            hacc[nf] = mma(a, b, hacc[nf]);
                        }
                    }
                    const int hw = p.H * p.Wd;
                    const int m = m0 + q * HEAD_ROWS + wv * 16 + e_kg * 4;  // first of the lane's four pixels
                    if (m < p.M) {
                        const int b_img = m / hw, r = m - b_img * hw;
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            const int n = nf * 16 + e_row;
                            if (n < p.head_n) {
                                const f32x4 v = hacc[nf] + *reinterpret_cast<const float*>(wfl + HEAD_B_ROW * ROWB + n * 4);
                                *reinterpret_cast<f32x4*>(p.head_out + (((size_t)b_img * p.head_n + n) * 4 + z) * hw + r) = v;
                            }
                        }
                    }
                }
                wait_vm_lgkm<63>();
                __builtin_amdgcn_s_barrier();  // the staging rows are reused by the next pass / the next tile
            }
            continue;  // (no drain: the K loop's waits are vmcnt(0), the logit stores retire under the next tile's first half-step)
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (rg == h) {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const int nl = cg * (BN / 4) + cf * 16 + e_kg * 4;
                    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n0 + nl);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) {
                        f32x4 v = acc[cf][rf] + bv;
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        const int ml = rf * 16 + e_row;
                        const int byte = nl * 2;
                        *reinterpret_cast<bf16x4*>(cst + ml * ROWB + ((((byte >> 4) ^ (ml & 7)) << 4) | (byte & 15))) = ov;
                    }
                }
            }
            wait_vm_lgkm<63>();  // LDS writes only: the DMA of the next tile stays in flight
            __builtin_amdgcn_s_barrier();
            {
            const int cl = tid_e % LPR, rl = tid_e / LPR;
            if (rl < RPP) {
                for (int r0 = 0; r0 < BM / 2; r0 += RPP) {
                    const int ml = r0 + rl;
                    const int m = m0 + h * (BM / 2) + ml;
                    if (ml >= BM / 2 || m >= p.M) continue;
                    size_t orow = m;
                    if (GATHER == G_DECONV) {  // phase-interleaved output pixel (2y+py, 2x+px) of a (2H, 2W) map
                        const int hw = p.H * p.Wd;
                        const int b = m / hw, r = m - b * hw;
                        const int y = r / p.Wd, x = r - y * p.Wd;
                        const int py = p.py < 0 ? (z >> 1) : p.py, px = p.py < 0 ? (z & 1) : p.px;
                        orow = ((size_t)b * (2 * p.H) + 2 * y + py) * (2 * p.Wd) + 2 * x + px;
                    }
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(cst + ml * ROWB + ((cl ^ (ml & 7)) << 4));
                    *reinterpret_cast<u32x4*>(Cb + orow * p.ldc + n0 + cl * 8) = raw;
                }
            }
            }
            wait_vm_lgkm<63>();
            __builtin_amdgcn_s_barrier();  // the staging region is reused by the other row half / the next tile
        }
        // Stores and loads both count in vmcnt but may retire out of order with respect to each other: drain them
        // before the counted waits of the next tile rely on the count again.
        if (!(DBG & 2048)) wait_vm_lgkm<0>();
    }
    if ((DBG & 256) && blockIdx.x == 0 && tid == 0) {  // (first 16 bytes of the output)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.C);
        o[0] = __builtin_readcyclecounter() - dbg_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
}

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace panel

bool panel_gemm_supported(const GemmParams& p, int prec, int groups) {
    const bool partials = p.gather == G_CONV3 && p.ksplit > 1 && !p.out_bf16 && !p.bias;  // fp32 split-K partial sums
    if (prec != PP_PREC_BF16 || (!p.out_bf16 && !partials) || p.residual || p.planar_P > 0) return false;
    if (p.ksplit > 1 && !partials) return false;
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    if (p.Cin % 32 != 0 || p.K % 128 != 0 || p.ldc % 8 != 0) return false;
    int BM, BN;
    if (p.gather == G_DECONV) { BM = 192; BN = 256; }
    else if (p.gather == G_CONV3) { BM = 256; BN = 192; }
    else return false;
    if (p.N % BN != 0) return false;
    // one workgroup per CU: with fewer tiles than that the 128 x 128 kernel spreads the work better
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    return ntiles >= 192;
}

int panel_gemm(const GemmParams& p_in, int groups, hipStream_t s) {
    using namespace panel;
    GemmParams p = p_in;
    p.groups = groups;
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < OOB && p.w_bytes < OOB, PP_ERR_UNSUPPORTED,
               "pp panel gemm: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    void (*kern)(const GemmParams) = nullptr;
    int BM, BN;
    if (p.gather == G_DECONV) {
        kern = panel_gemm_kernel<G_DECONV, 6, 4>;
        BM = 192;
        BN = 256;
    } else {
        kern = panel_gemm_kernel<G_CONV3, 8, 3>;
        BM = 256;
        BN = 192;
    }
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    PP_REQUIRE(ntiles < (1ll << 30), PP_ERR_UNSUPPORTED, "pp panel gemm: too many output tiles");
    int slots = device_cus();  // one workgroup per CU; a multiple of 8 keeps a workgroup's tiles on one XCD
    slots -= slots % 8;
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp
