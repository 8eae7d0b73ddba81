// Winograd F(2x2, 3x3) form of the first stage of the four scalar towers in the parity precision (PP_PREC_F16X3):
//     Conv2d(C -> C, k3, p1) + BatchNorm (folded) -> MaxPool2d(4, 3) -> ReLU        (probmap_head.py:261-294)
// on H x W feature maps with H % 4 == 0 and W % 6 == 0 (16 x 12 at ViT-S 256x192, 24 x 18 at ViT-B 384x288), four towers sharing the input. As an implicit GEMM (pp_panel_split.hip, POOL form) this stage is
// 261 GFLOP at bs 64 with flip test = 15 % of the path's FLOPs and 0.65 - 0.73 ms of a 5.2 ms step; every algorithmic product
// costs three fp16 MFMAs. Winograd's minimal filtering computes a 2 x 2 output tile from a 4 x 4 input tile with 16
// multiplications per (input channel, output channel) instead of 36:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A                                          (Lavin & Gray 2016, F(2x2, 3x3))
// i.e. 16 GEMMs  M_p[tile, o] = sum_c V_p[tile, c] U_p[o, c]  over the 48 tiles of every image - 2.25x fewer MFMAs.
//
//   * pp::wino::input_transform_kernel: V_p = (B^T d B)_p of every tile, in fp32 on the exact values of the split operands
//     (hi + lo is exact in fp32; B has entries 0, +-1 only), re-split and stored as 16 planes [p][tile][C] - the A operand of
//     the GEMMs (151 MB at bs 64; L2 / MALL resident for the kernel that follows);
//   * U_p = (G g G^T)_p of the BN-folded weights comes pre-computed in fp64 and split (weights.py: tower0.wino);
//   * tiles are numbered GROUP-major: a group = 2 x 3 tiles = 4 x 6 output pixels = two windows of the MaxPool2d(4, 3) that
//     follows, so that any run of 6 n tiles pools on its own;
//   * pp::wino::gemm_pool_kernel: a workgroup owns 192 tiles (32 groups; at 16 x 12 four whole images) x 96 output
//     channels and walks the 16 positions x C / 32 K-steps as ONE stream of 36 KiB stages (192 V rows + 96 U rows, LDS-DMA)
//     on a ring of four (three in flight); 8 waves x (48 tiles x 48 channels), the stage loop of pp_panel_split.hip (hi x hi
//     while the lo fragments arrive, one barrier, lo x hi and hi x lo while the next stage's hi fragments replace the dying
//     ones). The position's sums live in 36 accumulator registers; at the end of a position they are folded into the four
//     output accumulators of the 2 x 2 tile with the coefficients of A^T (0, +-1): 180 accumulator registers per lane, one
//     workgroup per CU. Epilogue, one wave row group (48 tiles = 8 groups) at a time: the 8 x (4 x 6) x 96 outputs go to LDS as
//     fp32, are max-pooled (4, 3), get bias + ReLU and leave as the split format - only the pooled map is stored, as in the
//     POOL form.
//
// Numerics: the transforms add at most four values (input) / nine values (output) in fp32; measured against torch fp64 on the
// unrounded operands the pooled outputs agree to the same 2e-5 the implicit-GEMM kernels meet (tests/test_split_fp16.py).
#include "pp_common.h"
#include "pp_split.h"

namespace pp {
namespace wino {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef WINO_DBG
#define WINO_DBG 0  // dev ablations (timing only, wrong results): 2 no MFMAs, 4 no DMA traffic, 8 no epilogue
#endif
constexpr int DBG = WINO_DBG;

constexpr int PH = 4, PW = 3;            // MaxPool2d(4, 3)
constexpr int GT = 6, GPX = 24;          // a group: 2 x 3 tiles of 2 x 2 outputs = 4 x 6 pixels = two pooling windows
constexpr int BT = 192, BN = 96, THREADS = 512;  // 32 groups x 96 channels per workgroup; a wave row group = 48 tiles = 8 groups
constexpr int STAGE = (BT + BN) * 128;  // 36 KiB: one 128-byte block of K per row
constexpr int NST = 4;
constexpr int LDS = NST * STAGE;        // 144 KiB
constexpr int PITCH = 100;              // floats per pixel of the epilogue staging (16-byte aligned, 2-way conflicts at worst)
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(8 * GPX * PITCH * 4 <= LDS, "epilogue staging fits the ring");

struct Params {
    const void* V;      // [16][T][Cin] split: transformed input tiles
    const void* U;      // [groups][16][Cout][Cin] split: transformed weights
    const float* bias;  // [groups][Cout] (folded BatchNorm shift)
    void* out;          // [groups][nb][H / 4][W / 3][Cout] split: pooled, ReLU'd
    int nb, T, Cin, Cout, groups;  // T = nb * (H / 2) * (W / 2) tiles
    int H, W;
    int order;          // tile order (see the kernel)
    unsigned v_bytes, u_bytes;
};

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (DBG & 2) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// ---------------------------------------------------------------------------------------------------------------------
// V_p[tile][c] = (B^T d B)[a][b], p = 4 a + b, d = the 4 x 4 input patch at rows 2 ty - 1 .., columns 2 tx - 1 .. (zero outside
// the image: the convolution's padding), B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]. One thread per (tile, four channels).
__global__ __launch_bounds__(256) void input_transform_kernel(const char* __restrict__ feat, char* __restrict__ V, int nb, int C, int H, int W) {
    const int G4 = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int GX = W / 6, GY = H / 4;
    const long long T = (long long)nb * GY * GX * GT;
    if (idx >= T * G4) return;
    const int g4 = (int)(idx % G4);
    const long long tile = idx / G4;
    // group-major tile index -> (image, tile row, tile column)
    const int tl = (int)(tile % GT);
    const long long grp = tile / GT;
    const int gx = (int)(grp % GX), gy = (int)((grp / GX) % GY), img = (int)(grp / ((long long)GX * GY));
    const int ty = 2 * gy + tl / 3, tx = 3 * gx + tl % 3;
    const int boff = (g4 >> 3) * 128 + (g4 & 7) * 8;  // hi halves of the four channels inside a pixel row; lo: + 64
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const char* src = feat + ((size_t)(img * H + y) * W + x) * (size_t)(C * 4) + boff;
                const f16x4 h = *reinterpret_cast<const f16x4*>(src), l = *reinterpret_cast<const f16x4*>(src + 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + (float)l[j];
            }
            d[r][c] = v;
        }
    f32x4 tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // B^T d
        tt[0][c] = d[0][c] - d[2][c];
        tt[1][c] = d[1][c] + d[2][c];
        tt[2][c] = d[2][c] - d[1][c];
        tt[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {  // (B^T d) B
        f32x4 o[4];
        o[0] = tt[a][0] - tt[a][2];
        o[1] = tt[a][1] + tt[a][2];
        o[2] = tt[a][2] - tt[a][1];
        o[3] = tt[a][1] - tt[a][3];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = split_hi(o[b][j]);
                l[j] = split_lo(o[b][j], h[j]);
            }
            char* dst = V + ((size_t)(4 * a + b) * (size_t)T + (size_t)tile) * (size_t)(C * 4) + boff;
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + 64) = l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS, 2) void gemm_pool_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv >> 2, rg = wv & 3;  // column half (48 of the 96 channels), row quarter = image of the block
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = fr & 7;

    // Tile order. The hardware deals workgroup ids round-robin to the eight XCDs; id -> (id & 7) * (n / 8) + (id >> 3) gives every XCD
    // a contiguous run of logical ids, 32 of them resident at a time. A run is cut into SUPER TILES of 4 row blocks x 8 column tiles:
    // per stage the 32 workgroups of an XCD then pull 4 x 24 KiB of V rows and 8 x 12 KiB of U rows through their L2 (192 KiB;
    // 2 x 16 - column tiles fastest - is 240 KiB, and the HBM-side fetch of the launch was 1.19 GB against 0.6 GB of unique bytes
    // per XCD round: WINO_ORDER 0).
    const int CT = p.groups * (p.Cout / BN);  // column tiles
    const int RB = (p.T + BT - 1) / BT;       // row blocks
    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
    int ct, rb;
    const int sc = p.order;  // column tiles per super tile (a power of two <= 32 dividing CT), 0: column tiles fastest over the whole run
    if (sc > 0) {
        const int sr = 32 / sc, st = id >> 5, w = id & 31, sct = st % (CT / sc), srb = st / (CT / sc);
        rb = sr * srb + w / sc;
        ct = sc * sct + w % sc;
    } else {
        ct = id % CT;
        rb = id / CT;
    }
    if (rb >= RB) return;  // (the grid is padded to whole super tiles)
    const int g = (ct * BN) / p.Cout, n0 = ct * BN - g * p.Cout;
    const int KB = p.Cin >> 5;       // K-steps per position (a multiple of four)
    const int nstages = 16 * KB;

    // ---- DMA: 36 pieces of 8 lines per stage: 24 V pieces (wave w: pieces w, w + 8, w + 16), 12 U pieces (24 + w; waves 0-3 also
    // 32 + w). Lane (line l = lane >> 3, physical chunk pc = lane & 7) fetches logical chunk pc ^ l: chunk-swizzled by (line & 7).
    const int d_l = lane >> 3;
    const unsigned d_sw = (unsigned)(((lane & 7) ^ d_l) << 4);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.V), 0, (DBG & 4) ? 0u : p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.U), 0, (DBG & 4) ? 0u : p.u_bytes, 0x00020000);
    const unsigned rowb = (unsigned)(p.Cin * 4);
    const unsigned v_v = (unsigned)(rb * BT + 8 * wv + d_l) * rowb + d_sw;                                   // + 64 j tile rows
    const unsigned v_u0 = (unsigned)((g * 16) * p.Cout + n0 + 8 * wv + d_l) * rowb + d_sw;                   // piece 24 + w: rows 8 w ..
    const unsigned v_u1 = (unsigned)((g * 16) * p.Cout + n0 + 64 + 8 * (wv & 3) + d_l) * rowb + d_sw;        // piece 32 + w (w < 4)
    const unsigned v_pstride = (unsigned)p.T * rowb, u_pstride = (unsigned)p.Cout * rowb;
    int i_s = 0, i_p = 0, i_kb = 0;  // issue cursor: stage, its position and K-step
    auto issue_stage = [&]() {
        char* dst = smem + (i_s & (NST - 1)) * STAGE + wv * 1024;
        const bool live = i_s < nstages;  // past the end: zero fillers (every wave's vmcnt arithmetic stays the same)
        const unsigned so_v = live ? (unsigned)i_p * v_pstride + (unsigned)(i_kb * 128) : OOB;
        const unsigned so_u = live ? (unsigned)i_p * u_pstride + (unsigned)(i_kb * 128) : OOB;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_ptr_t)(dst + j * 8192), 16, live ? v_v + (unsigned)(j * 64) * rowb : OOB, live ? so_v : 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, (lds_ptr_t)(dst + 24 * 1024), 16, live ? v_u0 : OOB, live ? so_u : 0, 0, 0);
        if (wv < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, (lds_ptr_t)(dst + 32 * 1024), 16, live ? v_u1 : OOB, live ? so_u : 0, 0, 0);
        ++i_s;
        if (++i_kb == KB) {
            i_kb = 0;
            ++i_p;
        }
    };

    // ---- fragment reads: hi halves in 16-byte chunk fg, lo halves in chunk 4 + fg of a line
    const int a_off = (48 * rg + fr) * 128, w_off = BT * 128 + (48 * cg + fr) * 128;
    const int ch_hi = (fg ^ sw) << 4, ch_lo = ((4 + fg) ^ sw) << 4;
    auto frag_a = [&](int buf, int lo, int rf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + a_off + rf * 2048 + (lo ? ch_lo : ch_hi));
    };
    auto frag_w = [&](int buf, int lo, int cf) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(smem + buf * STAGE + w_off + cf * 2048 + (lo ? ch_lo : ch_hi));
    };

    // acc[cf][rf]: A operand = U rows (channels), B operand = V rows (tiles): lane = tile 48 rg + 16 rf + fr, channels
    // 48 cg + 16 cf + 4 fg + (0..3)
    f32x4 tmp[3][3], Y[4][3][3];
#pragma unroll
    for (int cf = 0; cf < 3; ++cf)
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            tmp[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 4; ++o) Y[o][cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int s = 0; s < NST; ++s) issue_stage();
    if (wv < 4) wait_vm_lgkm<15>(); else wait_vm_lgkm<12>();  // the first stage has landed (three younger ones may fly)
    __builtin_amdgcn_s_barrier();
    u32x4 ah[3], wh[3], al[3], wl[3];
#pragma unroll
    for (int cf = 0; cf < 3; ++cf) wh[cf] = frag_w(0, 0, cf);
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) ah[rf] = frag_a(0, 0, rf);

#pragma unroll 1
    for (int pos = 0; pos < 16; ++pos) {
#pragma unroll 1
        for (int kq = 0; kq < KB; kq += NST) {
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int cb = u, nb = (u + 1) & (NST - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cf = 0; cf < 3; ++cf) wl[cf] = frag_w(cb, 1, cf);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) al[rf] = frag_a(cb, 1, rf);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                    for (int cf = 0; cf < 3; ++cf) tmp[cf][rf] = mma(wh[cf], ah[rf], tmp[cf][rf]);
                __builtin_amdgcn_sched_barrier(0);
                // every wave holds the rest of this stage in registers -> its buffer is free; the next stage has landed when only this
                // wave's pieces of the two stages after it are outstanding
                if (wv < 4) wait_vm_lgkm<10>(); else wait_vm_lgkm<8>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                issue_stage();
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
                    for (int cf = 0; cf < 3; ++cf) tmp[cf][rf] = mma(wl[cf], ah[rf], tmp[cf][rf]);
                    ah[rf] = frag_a(nb, 0, rf);
                }
#pragma unroll
                for (int cf = 0; cf < 3; ++cf) {
#pragma unroll
                    for (int rf = 0; rf < 3; ++rf) tmp[cf][rf] = mma(wh[cf], al[rf], tmp[cf][rf]);
                    wh[cf] = frag_w(nb, 0, cf);
                }
            }
        }
        // ---- fold M_p into the 2 x 2 outputs: Y[i][j] += A^T[i][a] A^T[j][b] M_(a, b), A^T = [1 1 1 0; 0 1 -1 -1]
        const int a = pos >> 2, b = pos & 3;
        const float ca0 = a < 3 ? 1.f : 0.f, ca1 = a == 0 ? 0.f : (a == 1 ? 1.f : -1.f);
        const float cb0 = b < 3 ? 1.f : 0.f, cb1 = b == 0 ? 0.f : (b == 1 ? 1.f : -1.f);
        const float c4[4] = {ca0 * cb0, ca0 * cb1, ca1 * cb0, ca1 * cb1};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (c4[o] != 0.f) {
#pragma unroll
                for (int cf = 0; cf < 3; ++cf)
#pragma unroll
                    for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) Y[o][cf][rf][q] = __builtin_fmaf(tmp[cf][rf][q], c4[o], Y[o][cf][rf][q]);
                    }
            }
        }
#pragma unroll
        for (int cf = 0; cf < 3; ++cf)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) tmp[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- epilogue, image by image (image i = the tiles of the waves with rg == i): outputs to LDS as fp32 [pixel][PITCH],
    // MaxPool2d(4, 3), + bias, ReLU, split, out. The fillers issued past the last stage must have landed before the ring is reused.
    wait_vm_lgkm<0>();
    __syncthreads();
    if (DBG & 8) return;
    float* stage = reinterpret_cast<float*>(smem);
    const int GX = p.W / 6, GY = p.H / 4, OW = 2 * GX;
    const int ngroups = p.nb * GY * GX;  // (< 2^20: host check; the group index is decoded with float reciprocals, exact in that range -
    // a 64-bit vector division here needs two dozen temporaries while the 144 output accumulators are still live, and the register
    // allocator answered by keeping six of them in scratch for the WHOLE kernel: reloads inside the stage loop, 350 -> 447 us)
    const float r_gx = 1.0f / (float)GX, r_gy = 1.0f / (float)GY;
    // (lane / thread ids re-derived behind an opaque copy: everything computed from them below would otherwise be hoisted above the
    // main loop, where all 256 registers are taken - 29 spilled, scratch reloads inside the stage loop, 350 -> 447 us)
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const int fr2 = tid2 & 15, fg2 = (tid2 >> 4) & 3;
    const int pp_ = tid2 / 12, cq = tid2 - pp_ * 12;  // pooling: thread -> (pooled pixel = (group, window) of the row group, eight channels); 192 of the 512 threads
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    if (tid2 < 16 * 12 && p.bias) {
        bv0 = *reinterpret_cast<const f32x4*>(p.bias + (size_t)g * p.Cout + n0 + 8 * cq);
        bv1 = *reinterpret_cast<const f32x4*>(p.bias + (size_t)g * p.Cout + n0 + 8 * cq + 4);
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        if (rg == i) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                const int t = 16 * rf + fr2, grp = t / GT, tl = t - grp * GT, tyl = tl / 3, txl = tl - tyl * 3;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int pix = grp * GPX + (2 * tyl + (o >> 1)) * 6 + 2 * txl + (o & 1);
#pragma unroll
                    for (int cf = 0; cf < 3; ++cf)
                        *reinterpret_cast<f32x4*>(stage + pix * PITCH + 48 * cg + 16 * cf + 4 * fg2) = Y[o][cf][rf];
                }
            }
        }
        __syncthreads();
        const int grp = pp_ >> 1, wnd = pp_ & 1;
        const int gidx = rb * (BT / GT) + i * 8 + grp;  // global group: (image, group row, group column)
        if (tid2 < 16 * 12 && gidx < ngroups) {
            const int q1 = (int)(((float)gidx + 0.5f) * r_gx), gx = gidx - q1 * GX;  // gidx / GX, gidx % GX
            const int img = (int)(((float)q1 + 0.5f) * r_gy), gy = q1 - img * GY;      // (gidx / GX) / GY, % GY
            const float ninf = -__builtin_inff();
            f32x4 m0 = {ninf, ninf, ninf, ninf}, m1 = m0;
#pragma unroll
            for (int y = 0; y < PH; ++y)
#pragma unroll
                for (int x = 0; x < PW; ++x) {
                    const float* s = stage + (grp * GPX + y * 6 + wnd * PW + x) * PITCH + 8 * cq;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        m0[q] = fmaxf(m0[q], v0[q]);
                        m1[q] = fmaxf(m1[q], v1[q]);
                    }
                }
            f16x8 hv, lv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // max(x) + b == max(x + b): the bias is per channel
                const float r0 = fmaxf(m0[q] + bv0[q], 0.f), r1 = fmaxf(m1[q] + bv1[q], 0.f);
                hv[q] = split_hi(r0);
                lv[q] = split_lo(r0, hv[q]);
                hv[4 + q] = split_hi(r1);
                lv[4 + q] = split_lo(r1, hv[4 + q]);
            }
            const size_t eoff = ((((size_t)g * p.nb + img) * GY + gy) * OW + 2 * gx + wnd) * (size_t)p.Cout + n0 + 8 * cq;
            char* o = split_addr(p.out, eoff);
            *reinterpret_cast<f16x8*>(o) = hv;
            *reinterpret_cast<f16x8*>(o + 64) = lv;
        }
        __syncthreads();
    }
}

}  // namespace wino
}  // namespace pp

extern "C" long long pp_winograd_scratch_bytes(int B, int H, int W, int Cin) {
    using namespace pp;
    if (B <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 6 != 0 || Cin <= 0 || Cin % 128 != 0) return PP_ERR_UNSUPPORTED;
    return 16ll * B * (H / 2) * (W / 2) * Cin * 4;
}

extern "C" int pp_conv3x3_winograd_maxpool_relu(const void* act_nhwc, const void* u_packed, const float* bias, void* v_scratch,
                                                void* out_pooled, int B, int H, int W, int Cin, int Cout, int pool_h, int pool_w,
                                                int groups, void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && u_packed && v_scratch && out_pooled, PP_ERR_INVALID_ARG, "pp_conv3x3_winograd_maxpool_relu: NULL argument");
    PP_REQUIRE(B > 0 && groups >= 1, PP_ERR_INVALID_ARG, "pp_conv3x3_winograd_maxpool_relu: bad B / groups");
    PP_REQUIRE(H > 0 && W > 0 && H % 4 == 0 && W % 6 == 0 && pool_h == wino::PH && pool_w == wino::PW, PP_ERR_UNSUPPORTED,
               "pp_conv3x3_winograd_maxpool_relu: built for feature maps with H % 4 == 0, W % 6 == 0 pooled (4, 3) (16 x 12, 24 x 18)");
    PP_REQUIRE(Cin % 128 == 0 && Cout % wino::BN == 0, PP_ERR_UNSUPPORTED,
               "pp_conv3x3_winograd_maxpool_relu: Cin must be a multiple of 128, Cout of 96");
    const long long T = (long long)B * (H / 2) * (W / 2);
    const long long vb = 16ll * T * Cin * 4, ub = 16ll * groups * Cout * (long long)Cin * 4;
    PP_REQUIRE(vb < wino::OOB && ub < wino::OOB && T / wino::GT < (1 << 20), PP_ERR_UNSUPPORTED,
               "pp_conv3x3_winograd_maxpool_relu: operands must be smaller than 2 GiB (and fewer than 2^20 tile groups)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long items = T * (Cin / 4);
    hipLaunchKernelGGL(wino::input_transform_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const char*>(act_nhwc), reinterpret_cast<char*>(v_scratch), B, Cin, H, W);
    PP_LAUNCH_CHECK_AS("winograd_input_transform");
    wino::Params p{};
    p.V = v_scratch;
    p.U = u_packed;
    p.bias = bias;
    p.out = out_pooled;
    p.nb = B;
    p.T = (int)T;
    p.Cin = Cin;
    p.Cout = Cout;
    p.groups = groups;
    p.H = H;
    p.W = W;
    p.v_bytes = (unsigned)vb;
    p.u_bytes = (unsigned)ub;
    const long long ct_ = (long long)groups * (Cout / wino::BN), rb_ = (T + wino::BT - 1) / wino::BT;
    int order = option("wino_order");
    if (order < 0 || order > 32 || (order & (order - 1)) != 0 || (order > 0 && ct_ % order != 0)) order = 0;
    p.order = order;
    const long long sr_ = order > 0 ? 32 / order : 1;
    const long long grid = ((rb_ + sr_ - 1) / sr_) * sr_ * ct_;  // whole super tiles (see the kernel)
    PP_REQUIRE(grid < (1ll << 30), PP_ERR_UNSUPPORTED, "pp_conv3x3_winograd_maxpool_relu: too many tiles");
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino::gemm_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, wino::LDS));
    hipLaunchKernelGGL(wino::gemm_pool_kernel, dim3((unsigned)grid), dim3(wino::THREADS), wino::LDS, s, p);
    PP_LAUNCH_CHECK_AS("winograd_gemm_pool");
    return PP_OK;
}
