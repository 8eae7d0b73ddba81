// Halo-staged 3x3 convolution for gfx950 (bf16 operands, bf16 output): the first tower convolution of ProbMapHead
// (Conv2d 3x3 pad 1 + folded BN on the 16 x 12 ViT feature map, four towers per launch, probmap_head.py:261-294) - 261
// of the path's 1720 GFLOP at bs 64.
//
// Why a second kernel for this shape. pp_panel_gemm.hip runs the convolution as an implicit GEMM: every K-step of 64
// channels x one tap fetches a fresh activation tile, i.e. the SAME pixels nine times, shifted. Its ablations
// (scripts/micro/panel_ablate.sh) show the L2 -> LDS fill is what the launch waits for: 264 us as is, 196 us with the
// activation DMA out of bounds, 161 us with no DMA traffic at all. Here the activations of a tile are staged ONCE per
// 64-channel chunk, with their zero border, and the nine taps read them at shifted LDS addresses:
//
//   * tile = two whole images (384 pixels) x 128 output channels; per chunk the activation "halo image" is
//     2 x (H + 2) x (W + 2) = 504 rows x 128 B = 63 KiB (border rows are out-of-bounds DMA = zeros), double-buffered;
//     the weights stream as one 16 KiB stage (128 rows x 64 k) per tap, two stages. Fill per nine K-steps:
//     63 + 9 x 16 = 207 KiB against 9 x 56 = 504 KiB;
//   * tap (dy, dx) of pixel (y, x) is halo row (y + 1 + dy) (W + 2) + (x + 1 + dx) - a per-tap constant added to the
//     lane's row address. LDS rows are 128 B with the 16-byte chunk XOR-swizzled by ((y' W + x') & 7) of the row's image
//     coordinates (y', x') - for the 16 consecutive pixels of an MFMA fragment that key runs through consecutive
//     values for EVERY tap (also across image-row wraps and into the border), so the reads stay conflict-free like
//     the contiguous tiles of the other kernels; row parity alternates the same way (the wrap adds 3 rows);
//   * everything else as in pp_panel_gemm.hip: 512 threads = 8 waves (row quarter rg, column half cg), 96 x 64 wave
//     tiles (6 x 4 MFMA 16x16x32 fragments), fragments double-buffered in registers at half-step granularity, one
//     barrier per K-step, persistent workgroups (one per CU) whose DMA cursors run on across tiles, bias / ReLU /
//     bf16 epilogue staged through the activation buffer that was just consumed and stored as whole rows.
#include "pp_common.h"
#include "pp_gemm.h"
#include <cstdlib>

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace halo {

#ifndef HALO_DBG
#define HALO_DBG 0
#endif
constexpr int DBG = HALO_DBG;  // dev ablations: 1 DMA out of bounds, 4 no MFMA, 256 clock probe (first 16 bytes of the output)

constexpr int THREADS = 512;
constexpr int IMG_PIX = 192;                 // pixels of one image (H * W)
constexpr int HALO_ROWS = 252;               // (H + 2) * (W + 2) for 16 x 12 and 12 x 16
constexpr int BM = 2 * IMG_PIX, BN = 128, RF = 6, CF = 4;
constexpr int A_BUF = 2 * HALO_ROWS * 128;   // 63 KiB
constexpr int A_INSTR = A_BUF / 1024;        // 63 DMA instructions per chunk
constexpr int W_STAGE = BN * 128;            // 16 KiB
constexpr int OFF_W = 2 * A_BUF;
constexpr int OFF_DUMP = OFF_W + 2 * W_STAGE; // 1 KiB that takes the DMA instructions which have nothing to fetch
constexpr int LDS = OFF_DUMP + 1024;         // 159 KiB
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS <= 160 * 1024, "LDS map");
static_assert((BM / 2) * BN * 2 <= A_BUF, "one image of the bf16 output tile is staged through an activation buffer");

#define HSGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x008, SG_VMEM = 0x010, SG_DS_READ = 0x100;

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// POOL: the towers' MaxPool2d(kernel = stride = (pool_h, pool_w)) + ReLU (probmap_head.py:264,277-278) applied to the staged image in the
// epilogue - a tile holds whole images, so the pooling windows never leave it; C is then the pooled tensor (N, H / ph, W / pw, Cout)
template <bool POOL>
__global__ __launch_bounds__(THREADS, 2) void conv3_halo_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 1, cg = wv & 1;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int Wd = p.Wd, H = p.H, PW = Wd + 2, HR = (H + 2) * PW;

    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntn * ntm * p.groups;
    const int c64 = p.Cin / 64;
    const int nsteps = 9 * c64;  // even (Cin % 128 == 0): K-step parity = weight stage
    if ((int)blockIdx.x >= ntiles) return;
    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if (DBG & 256) {  // shader-clock cycles against the 100 MHz reference over the launch -> the clock the kernel ran at
        dbg_t0 = __builtin_readcyclecounter();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }

    // tile order as in pp_panel_gemm.hip: a contiguous run of the list (row panel -> group -> column tile) per XCD
    auto decode_tile = [&](int t, int& z, int& m0, int& n0) {
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        n0 = (t % ntn) * BN;
        const int r = t / ntn;
        z = r % p.groups;
        m0 = (r / p.groups) * BM;
    };

    // ---- DMA lanes. An instruction moves 8 LDS rows of 128 B; lane = (row d_l, physical chunk pc).
    const int d_l = lane >> 3, pc = lane & 7;
    // activation halo rows: tile-invariant part of the address (pixel inside the image pair, swizzled chunk) or OOB
    unsigned a_rel[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int l = 8 * (wv + 8 * j) + d_l;
        const int img = l >= HR ? 1 : 0, r = l - img * HR;
        const int hy = r / PW - 1, hx = r - (r / PW) * PW - 1;
        const bool ok = l < 2 * HR && hy >= 0 && hy < H && hx >= 0 && hx < Wd;
        const int lin = hy * Wd + hx;
        a_rel[j] = ok ? (unsigned)((img * IMG_PIX + lin) * (p.Cin * 2) + ((pc ^ (lin & 7)) << 4)) : OOB;
    }
    // weight rows: stage row n = 8 (wv + 8 jj) + d_l
    unsigned w_rel[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int n = 8 * (wv + 8 * jj) + d_l;
        w_rel[jj] = (unsigned)n * (unsigned)(p.ldw * 2) + (unsigned)((pc ^ (n & 7)) << 4);
    }

    // weight cursor: two K-steps ahead of the MFMAs
    __amdgpu_buffer_rsrc_t w_rsrc, a_rsrc;
    int wi_tile = blockIdx.x, wi_chunk = 0, wi_tap = 0;
    unsigned wi_base = 0;
    bool wi_live = true;
    auto setup_w_tile = [&]() {
        int z = 0, m0 = 0, n0 = 0;
        wi_live = wi_tile < ntiles;
        if (wi_live) decode_tile(wi_tile, z, m0, n0);
        const char* Wt = reinterpret_cast<const char*>(p.W) + (size_t)z * p.strideW_z * 2;
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, p.w_bytes, 0x00020000);
        wi_base = (unsigned)n0 * (unsigned)(p.ldw * 2);
        wi_chunk = 0;
        wi_tap = 0;
    };
    auto issue_w = [&](int buf) {
        const unsigned kb = wi_base + (unsigned)((wi_tap * p.Cin + wi_chunk * 64) * 2);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            char* dst = smem + OFF_W + buf * W_STAGE + (wv + 8 * jj) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, (wi_live && !(DBG & 1)) ? w_rel[jj] + kb : OOB, 0, 0, 0);
        }
        if (++wi_tap == 9) {
            wi_tap = 0;
            if (++wi_chunk == c64) {
                wi_tile += gridDim.x;
                setup_w_tile();
            }
        }
    };
    // activation cursor: the chunk after the one being read, one DMA instruction per K-step (taps 0..7)
    int pa_tile = blockIdx.x, pa_chunk = 0;
    unsigned pa_base = 0;
    bool pa_live = true;
    auto setup_a_tile = [&]() {
        int z = 0, m0 = 0, n0 = 0;
        pa_live = pa_tile < ntiles;
        if (pa_live) decode_tile(pa_tile, z, m0, n0);
        const char* Act = reinterpret_cast<const char*>(p.A) + (size_t)z * p.strideA_z * 2;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Act), 0, p.a_bytes, 0x00020000);
        pa_base = (unsigned)m0 * (unsigned)(p.Cin * 2);  // images past the batch: beyond a_bytes = zeros
        pa_chunk = 0;
    };
    // instruction j (0..7, run-time in the K loop; j = 8 or the 64th instruction: nothing to fetch - zeros into the dump
    // slot, so that the second half of every K-step is one branch-free block)
    auto issue_a = [&](int abuf, int j) {
        unsigned rel = a_rel[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) {  // (a run-time register-array index goes to scratch - also when written as selects, unless laundered)
            rel = (j == q) ? a_rel[q] : rel;
            asm volatile("" : "+v"(rel));
        }
        const bool real = j < 8 && wv + 8 * j < A_INSTR;
        char* dst = smem + (real ? abuf * A_BUF + (wv + 8 * j) * 1024 : OFF_DUMP);
        const unsigned va = (real && pa_live && !(DBG & 1)) ? rel + pa_base + (unsigned)(pa_chunk * 128) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, va, 0, 0, 0);
    };
    auto advance_a = [&]() {
        if (++pa_chunk == c64) {
            pa_tile += gridDim.x;
            setup_a_tile();
        }
    };

    // ---- fragment reads
    // activation fragment rf of this wave: pixels (rg & 1) * 96 + rf * 16 + f_row of image rg >> 1
    int a_row_off[RF];  // byte offset of the centre tap's halo row
    const int p_lane = (rg & 1) * 96 + f_row;
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        const int px = p_lane + rf * 16;
        const int y = px / Wd, x = px - y * Wd;
        a_row_off[rf] = ((rg >> 1) * HR + (y + 1) * PW + (x + 1)) * 128;
    }
    const int w_frag_off = (cg * (BN / 2) + f_row) * 128;
    const int w_sw = f_row & 7;
    auto read_frags = [&](int abuf, int wbuf, int tap, int h, u32x4 (&af)[RF], u32x4 (&wf)[CF]) {
        const char* wbase = smem + OFF_W + wbuf * W_STAGE + w_frag_off + (((h * 4 + f_kg) ^ w_sw) << 4);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) wf[cf] = *reinterpret_cast<const u32x4*>(wbase + cf * 2048);
        const int t3 = (tap * 11) >> 5;  // tap / 3 for tap < 9
        const int dy = t3 - 1, dx = tap - 3 * t3 - 1;
        const int key = (p_lane + dy * Wd + dx) & 7;  // (rf * 16 does not change it)
        const char* abase = smem + abuf * A_BUF + (dy * PW + dx) * 128 + (((h * 4 + f_kg) ^ key) << 4);
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) af[rf] = *reinterpret_cast<const u32x4*>(abase + a_row_off[rf]);
    };

    // ---- prologue: chunk 0 of the first tile whole, two weight stages
    setup_a_tile();
    setup_w_tile();
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_a(0, j);
    advance_a();
    issue_w(0);
    issue_w(1);
    wait_vm_lgkm<2>();  // all but the second weight stage
    __builtin_amdgcn_s_barrier();
    u32x4 af[2][RF], wf[2][CF];
    int r_tap = 0, r_abuf = 0;
    read_frags(0, 0, 0, 0, af[0], wf[0]);

    auto mfmas = [&](f32x4 (&acc)[CF][RF], const u32x4 (&a)[RF], const u32x4 (&w)[CF]) {
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)
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
            for (int s = 0; s < 2; ++s) {  // K-step k2 + s reads weight stage s
                // ---- first half: fragments (step, h0) are in registers; read (step, h1) under their MFMAs
                __builtin_amdgcn_sched_barrier(0);
                read_frags(r_abuf, s, r_tap, 1, af[1], wf[1]);
                mfmas(acc, af[0], wf[0]);
                HSGB(SG_MFMA, 24 - 2 * (RF + CF) + 2);
#pragma unroll
                for (int i = 0; i < RF + CF; ++i) {
                    HSGB(SG_DS_READ, 1);
                    if (i < RF + CF - 1) HSGB(SG_MFMA, 2);
                }
                // ---- every wave holds the rest of this step in registers: weight stage s is free, the other stage (and
                // every activation instruction issued so far) has landed
                __builtin_amdgcn_sched_barrier(0);
                wait_vm_lgkm<0>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- second half: weight stage two steps ahead into buffer s, one instruction of the next activation
                // chunk into the other halo buffer, fragments (step + 1, h0)
                issue_w(s);
                issue_a(r_abuf ^ 1, r_tap);
                if (++r_tap == 9) {
                    r_tap = 0;
                    r_abuf ^= 1;
                    advance_a();
                }
                read_frags(r_abuf, s ^ 1, r_tap, 0, af[0], wf[0]);
                mfmas(acc, af[1], wf[1]);
                HSGB(SG_MFMA, 24 - 2 * (RF + CF) + 2);
                HSGB(SG_VMEM, 1);
#pragma unroll
                for (int i = 0; i < RF + CF; ++i) {
                    HSGB(SG_DS_READ, 1);
                    if (i < RF + CF - 1) HSGB(SG_MFMA, 2);
                    if (i < 2) HSGB(SG_VMEM, 1);
                }
            }
        }

        // ---- epilogue: + bias (folded BN), optional ReLU, bf16; one image (192 rows x 256 B) at a time through the halo
        // buffer of the chunk just consumed (r_abuf already names the NEXT tile's first chunk), then whole-row stores
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int e_row = lane_e & 15, e_kg = lane_e >> 4, tid_e = (tid & ~63) | lane_e;
        int z, m0, n0;
        decode_tile(tile, z, m0, n0);
        const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias_z : nullptr;
        __bf16* __restrict__ Cb = reinterpret_cast<__bf16*>(p.C) + (size_t)z * p.strideC_z;
        char* cst = smem + (r_abuf ^ 1) * A_BUF;
        constexpr int ROWB = BN * 2;        // 256 B per staged row
        constexpr int LPR = ROWB / 16;      // 16 lanes per row
        constexpr int RPP = THREADS / LPR;  // 32 rows per pass
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((rg >> 1) == h) {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const int nl = cg * (BN / 2) + cf * 16 + e_kg * 4;
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
                        const int ml = (rg & 1) * 96 + rf * 16 + e_row;
                        const int byte = nl * 2;
                        *reinterpret_cast<bf16x4*>(cst + ml * ROWB + ((((byte >> 4) ^ (ml & 7)) << 4) | (byte & 15))) = ov;
                    }
                }
            }
            wait_vm_lgkm<63>();  // LDS writes only: the DMA of the next tile stays in flight
            __builtin_amdgcn_s_barrier();
            const int cl = tid_e % LPR, rl = tid_e / LPR;
            if (POOL) {
                // one thread per (pooled pixel, 4 channels): max over the window's staged rows (bf16 -> fp32 is exact and the
                // rounding to bf16 was monotonic, so this equals pooling the stored tensor), then ReLU
                // (16 x 12 image, 4 x 3 windows - compile-time here: run-time divisors cost the K loop its registers)
                constexpr int PH = 4, PWD = 3, IW = 12, Ho = 4, Wo = 4;
                const int img = (m0 + h * (BM / 2)) / IMG_PIX;
                const int c8 = tid_e & 31, pr = tid_e >> 5;  // 8-byte piece of the 256-byte row, pooled pixel
                if (img * IMG_PIX < p.M) {
                    const int yo = pr >> 2, xo = pr & 3;
                    float mx[4] = {0.f, 0.f, 0.f, 0.f};  // ReLU folded into the start value
#pragma unroll
                    for (int i = 0; i < PH; ++i)
#pragma unroll
                        for (int j = 0; j < PWD; ++j) {
                            const int ml = (yo * PH + i) * IW + xo * PWD + j;
                            const uint2 raw = *reinterpret_cast<const uint2*>(cst + ml * ROWB + (((c8 >> 1) ^ (ml & 7)) << 4) + (c8 & 1) * 8);
                            mx[0] = fmaxf(mx[0], __builtin_bit_cast(float, raw.x << 16));
                            mx[1] = fmaxf(mx[1], __builtin_bit_cast(float, raw.x & 0xffff0000u));
                            mx[2] = fmaxf(mx[2], __builtin_bit_cast(float, raw.y << 16));
                            mx[3] = fmaxf(mx[3], __builtin_bit_cast(float, raw.y & 0xffff0000u));
                        }
                    uint2 o;
                    o.x = (__builtin_bit_cast(unsigned, mx[0]) >> 16) | (__builtin_bit_cast(unsigned, mx[1]) & 0xffff0000u);
                    o.y = (__builtin_bit_cast(unsigned, mx[2]) >> 16) | (__builtin_bit_cast(unsigned, mx[3]) & 0xffff0000u);
                    *reinterpret_cast<uint2*>(Cb + ((size_t)img * (Ho * Wo) + pr) * p.ldc + n0 + c8 * 4) = o;
                }
            } else {
            for (int r0 = 0; r0 < BM / 2; r0 += RPP) {
                const int ml = r0 + rl;
                const int m = m0 + h * (BM / 2) + ml;
                if (m >= p.M) continue;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(cst + ml * ROWB + ((cl ^ (ml & 7)) << 4));
                *reinterpret_cast<u32x4*>(Cb + (size_t)m * p.ldc + n0 + cl * 8) = raw;
            }
            }
            wait_vm_lgkm<63>();
            __builtin_amdgcn_s_barrier();  // the staging buffer is reused by the other image / the next chunk's DMA
        }
        // stores and loads both count in vmcnt and may retire out of order with respect to each other: drain
        wait_vm_lgkm<0>();
    }
    if ((DBG & 256) && blockIdx.x == 0 && tid == 0) {
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

static bool enabled() { return option("conv_halo") != 0; }  // dev A/B switch (pp_set_option): 0 = implicit-GEMM kernel of pp_panel_gemm.hip

}  // namespace halo

bool conv_halo_supported(const GemmParams& p, int prec, int groups) {
    using namespace halo;
    if (!enabled() || prec != PP_PREC_BF16 || p.gather != G_CONV3 || !p.out_bf16 || p.residual || p.planar_P > 0) return false;
    if (p.ksplit > 1 || (p.act != ACT_NONE && p.act != ACT_RELU)) return false;
    if (p.H * p.Wd != IMG_PIX || (p.H + 2) * (p.Wd + 2) != HALO_ROWS || p.M % IMG_PIX != 0) return false;
    if (p.Cin % 128 != 0 || p.K != 9 * p.Cin || p.N % BN != 0 || p.ldc % 8 != 0) return false;
    if (p.pool_h > 0 && (p.pool_h != 4 || p.pool_w != 3 || p.H != 16 || p.Wd != 12 || p.act != ACT_NONE)) return false;  // the pooled epilogue is built for this geometry
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    return ntiles >= 192;  // one workgroup per CU: below that the 128 x 128 kernel spreads the work better
}

int conv_halo(const GemmParams& p_in, int groups, hipStream_t s) {
    using namespace halo;
    GemmParams p = p_in;
    p.groups = groups;
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < 0x70000000u && p.w_bytes < 0x70000000u, PP_ERR_UNSUPPORTED,
               "pp conv halo: operand tensors must be smaller than 1.75 GiB (32-bit buffer offsets)");
    const long long ntiles = (long long)(p.N / BN) * ((p.M + BM - 1) / BM) * groups;
    PP_REQUIRE(ntiles < (1ll << 30), PP_ERR_UNSUPPORTED, "pp conv halo: too many output tiles");
    int slots = device_cus();
    slots -= slots % 8;
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    void (*kern)(const GemmParams) = p.pool_h > 0 ? conv3_halo_kernel<true> : conv3_halo_kernel<false>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace pp

// Conv2d 3x3 (+ folded BN bias) -> MaxPool2d(ph, pw) -> ReLU of the first tower stage in one launch where the halo kernel
// covers the shape; otherwise the two separate entry points through `scratch_full`.
extern "C" int pp_conv3x3_maxpool_relu(int prec, const void* act_nhwc, const void* weight, const float* bias, void* out_pooled,
                                       void* scratch_full, int B, int H, int W, int Cin, int Cout, int ph, int pw, int groups,
                                       long long stride_act_g, long long stride_w_g, long long stride_bias_g, int fmt, void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && out_pooled, PP_ERR_INVALID_ARG, "pp_conv3x3_maxpool_relu: act, weight and out must be non-NULL");
    PP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && groups >= 1 && ph > 0 && pw > 0, PP_ERR_INVALID_ARG,
               "pp_conv3x3_maxpool_relu: bad shape");
    PP_REQUIRE(H % ph == 0 && W % pw == 0, PP_ERR_UNSUPPORTED, "pp_conv3x3_maxpool_relu: the pooling windows must tile the image");
    const int Ho = H / ph, Wo = W / pw;
    if (prec == PP_PREC_BF16 && fmt == PP_OUT_BF16) {
        GemmParams p{};
        p.A = act_nhwc; p.W = weight; p.C = out_pooled; p.bias = bias;
        p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
        p.lda = Cin; p.ldw = p.K; p.ldc = Cout; p.ldres = Cout;
        p.H = H; p.Wd = W; p.Cin = Cin;
        p.act = ACT_NONE; p.out_bf16 = 1; p.gather = G_CONV3;
        p.pool_h = ph; p.pool_w = pw;
        const size_t ab = (size_t)B * H * W * Cin * 2, wb = (size_t)Cout * p.K * 2;
        if (ab < 0x70000000u && wb < 0x70000000u) {
            p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
            p.strideA_z = stride_act_g; p.strideW_z = stride_w_g; p.strideBias_z = stride_bias_g;
            p.strideC_z = (long long)B * Ho * Wo * Cout;
            if (conv_halo_supported(p, prec, groups)) return conv_halo(p, groups, reinterpret_cast<hipStream_t>(stream));
        }
    }
    if (prec == PP_PREC_F16X3 && fmt == PP_OUT_SPLIT && pp::option("panel") != 0 && pp::option("conv_pool_split") != 0) {
        // split-fp16: pooling in the epilogue of the wide-tile kernel (pp_panel_split.hip, POOL)
        GemmParams p{};
        p.A = act_nhwc; p.W = weight; p.C = out_pooled; p.bias = bias;
        p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
        p.lda = Cin; p.ldw = p.K; p.ldc = Cout; p.ldres = Cout;
        p.H = H; p.Wd = W; p.Cin = Cin;
        p.act = ACT_NONE; p.out_bf16 = 2; p.gather = G_CONV3;
        p.pool_h = ph; p.pool_w = pw;
        const size_t ab = (size_t)B * H * W * Cin * 4, wb = (size_t)Cout * p.K * 4;
        if (ab < 0x70000000u && wb < 0x70000000u) {
            p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
            p.strideA_z = stride_act_g; p.strideW_z = stride_w_g; p.strideBias_z = stride_bias_g;
            p.strideC_z = (long long)B * Ho * Wo * Cout;
            if (panel_split_supported(p, prec, groups)) return panel_split_gemm(p, prec, groups, reinterpret_cast<hipStream_t>(stream));
        }
    }
    PP_REQUIRE(scratch_full, PP_ERR_UNSUPPORTED,
               "pp_conv3x3_maxpool_relu: this shape / precision takes the two-launch path and needs scratch_full (groups, B, H, W, Cout)");
    const int st = pp_conv_gemm(prec, PP_CONV3X3, act_nhwc, weight, bias, scratch_full, B, H, W, Cin, Cout, 0, 0, groups, stride_act_g,
                                stride_w_g, (long long)B * H * W * Cout, stride_bias_g, Cout, PP_ACT_NONE, fmt, stream);
    if (st != PP_OK) return st;
    return pp_maxpool_relu_nhwc(scratch_full, fmt, out_pooled, fmt, groups * B, H, W, Cout, ph, pw, stream);
}
