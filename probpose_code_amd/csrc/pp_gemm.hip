// MFMA GEMM family for gfx950: every dense contraction of the path (ViT linears, patch embed,
// deconv phases, 3x3 tower convs, 1x1 convs) is  C[m, n] = sum_k Act[m, k] * Wt[n, k]  with
// a K-contiguous weight matrix Wt (the nn.Linear layout) and an activation operand that is
// either a plain row-major matrix or an implicit im2col view of an NHWC tensor (no im2col
// buffer is ever materialised for the convolutions).
//
//   * PERSISTENT workgroups: the grid is sized to the chip (2 workgroups per CU), each
//     workgroup walks a strided list of 128x128 output tiles. On this path K is short (384 -
//     1536), so a tile is only 6 - 24 K-steps: launching one workgroup per tile spends more
//     time on dispatch + prologue latency than on MFMAs (measured: ~5 us fixed per tile round
//     vs 1.3 us of matrix work at K = 384). The persistent loop prefetches the first K-tile of
//     the NEXT output tile while the current one runs its last MFMAs and its epilogue;
//   * 256 threads = 4 waves as 2(n) x 2(m), each wave 64x64 = 4x4 MFMA 16x16 fragments, fp32
//     accumulators; operand precision is a template parameter:
//       bf16 -> v_mfma_f32_16x16x32_bf16, BK = 64;  f32 -> v_mfma_f32_16x16x4_f32 (exact fp32
//     products), BK = 32;  split fp16 (pp_split.h: x = hi + lo, 32 hi halves | 32 lo halves per 128-byte block)
//     -> three v_mfma_f32_16x16x32_f16 per fragment pair (lo*hi + hi*lo + hi*hi), BK = 32. All have the same
//     128-byte-per-row LDS image and C/D layout;
//   * staging by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR round trip, no ds_write pass.
//     The DMA destination is lane-linear, so the bank-conflict swizzle (16-byte chunk index
//     XOR row & 7) is applied on the per-lane SOURCE address; out-of-range rows / conv padding
//     taps use an out-of-bounds buffer offset, which makes the DMA write zeros;
//   * 2 LDS buffers, next K-tile's DMA in flight under the current tile's MFMAs;
//   * the weight tile is the MFMA "A" operand and the activation tile the "B" operand, so a
//     lane ends up with 4 consecutive n for one m;
//   * epilogue: + bias[n], exact-erf GELU or ReLU in registers, then the C tile is staged
//     through the LDS buffer that was just consumed (XOR-swizzled, conflict-free) so every
//     output row leaves the CU as contiguous 16-byte lane stores, + fp32 residual / broadcast
//     pos_embed, fp32 or bf16 output, optional output-row remap (deconv phase interleave) or
//     planar (B, N, P) store for the 17-channel logits;
//   * tile order is XCD-aware: each XCD (private L2) gets a contiguous run of tiles, so the
//     n-tiles that share one activation panel hit the same L2.
#include "pp_common.h"
#include "pp_gemm.h"
#include "pp_split.h"

#include <cstdlib>

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 128, BN = 128;
constexpr int GEMM_THREADS = 256;
constexpr int ROW_BYTES = 128;                  // one K-tile row in LDS
constexpr int TILE_BYTES = BM * ROW_BYTES;      // 16 KiB per operand per buffer
constexpr int BUF_BYTES = 2 * TILE_BYTES;       // W tile + Act tile
constexpr unsigned OOB_OFFSET = 0x7ffffff0u;    // >= num_records of every tensor (< 2 GiB): the DMA writes zeros

template <typename T>
struct Prec;
template <>
struct Prec<__bf16> {
    static constexpr int BK = 64;  // elements per K-tile
};
template <>
struct Prec<float> {
    static constexpr int BK = 32;
};
template <>
struct Prec<SplitH> {
    static constexpr int BK = 32;  // one 128-byte block: 32 hi halves | 32 lo halves
};
enum { FMT_F32 = 0, FMT_BF16 = 1, FMT_SPLIT = 2 };  // GemmParams::out_bf16 carries this code (PP_OUT_*)

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// MFMA over one 16x16 fragment pair for a 16-byte chunk of K
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c, __bf16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c, float) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
    return c;
}

// Per-tile staging state of one lane: byte offsets of its 4 + 4 source rows (chunk swizzle folded in)
struct StageRows {
    unsigned a_voff[4], w_voff[4];
    int a_y[4], a_x[4];
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
};

template <typename T, int GATHER, int OUT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(const GemmParams p) {
    constexpr int BK = Prec<T>::BK;
    constexpr int ESZ = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | Act tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int f_row = lane & 15, f_kg = lane >> 4;

    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int tiles_per_group = ntn * ntm, ntiles = tiles_per_group * p.groups;
    const int nk = p.K / BK;

    // XCD-aware order: block b runs on XCD b % 8 and visits b, b + G, ... (G % 8 == 0 keeps it there);
    // logical tile = (t % 8) * (ntiles / 8) + t / 8 gives each XCD a contiguous run of the tile list.
    auto decode_tile = [&](int t, int& z, int& m0, int& n0) {
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        z = t / tiles_per_group;
        const int r = t - z * tiles_per_group;
        n0 = (r % ntn) * BN;
        m0 = (r / ntn) * BM;
    };

    // ---- LDS-DMA staging: one `buffer_load_dwordx4 ... lds` moves 64 lanes x 16 B = 8 tile rows; wave w
    // issues rows [32 w + 8 j, +8), j = 0..3, of both tiles. Lane i lands in row 8 j' + (i >> 3), slot
    // (i & 7) and must therefore fetch source chunk (i & 7) ^ (i >> 3)  (row & 7 == i >> 3).
    const int d_row = lane >> 3;
    const unsigned d_chunk_bytes = (unsigned)(((lane & 7) ^ d_row) << 4);
    StageRows sr;
    int st_py = p.py, st_px = p.px;  // deconv phase of the tile being staged (all-phases launch: group index)
    int st_tap0 = 0;                 // split-K conv: first tap of the K slice being staged
    const int problems = p.ksplit > 1 ? p.groups / p.ksplit : p.groups;
    auto setup_stage = [&](int z, int m0, int n0) {
        if (GATHER == G_DECONV && p.py < 0) {
            st_py = z >> 1;
            st_px = z & 1;
        }
        const int zp = p.ksplit > 1 ? z % problems : z, zs = p.ksplit > 1 ? z / problems : 0;
        if (GATHER != G_LINEAR) st_tap0 = zs * (p.K / p.Cin);
        const char* Act = reinterpret_cast<const char*>(p.A) + (size_t)zp * p.strideA_z * ESZ;
        const char* Wt = reinterpret_cast<const char*>(p.W) + ((size_t)zp * p.strideW_z + (size_t)zs * p.K) * ESZ;
        sr.a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Act), 0, p.a_bytes, 0x00020000);
        sr.w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, p.w_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 32 + j * 8 + d_row;
            const int n = n0 + r, m = m0 + r;
            sr.w_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.ldw * ESZ) + d_chunk_bytes : OOB_OFFSET;
            sr.a_y[j] = sr.a_x[j] = 0;
            if (GATHER == G_LINEAR) {
                sr.a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.lda * ESZ) + d_chunk_bytes : OOB_OFFSET;
            } else {
                const int hw = p.H * p.Wd;
                const int b = m / hw, rr = m - b * hw;
                sr.a_y[j] = m < p.M ? rr / p.Wd : -100000;  // tail rows fail every bounds test below
                sr.a_x[j] = rr - (rr / p.Wd) * p.Wd;
                sr.a_voff[j] = (unsigned)m * (unsigned)(p.Cin * ESZ) + d_chunk_bytes;  // NHWC pixel origin
            }
        }
    };
    auto stage = [&](int kt, int buf) {
        const int k0 = kt * BK;
        char* wdst = smem + buf * BUF_BYTES + wave * 4096;
        char* adst = wdst + TILE_BYTES;
        int dy = 0, dx = 0, c0 = k0;
        if (GATHER != G_LINEAR) {
            const int tap_local = k0 / p.Cin;
            const int tap = st_tap0 + tap_local;
            c0 = k0 - tap_local * p.Cin;
            if (GATHER == G_CONV3) {  // 3x3, pad 1: tap = ky*3 + kx reads (y + ky - 1, x + kx - 1)
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            } else {  // deconv k4 s2 p1, output phase (py, px): tap = ty*2 + tx reads (y + ty - 1 + py, x + tx - 1 + px)
                dy = (tap >> 1) - 1 + st_py;
                dx = (tap & 1) - 1 + st_px;
            }
        }
        const unsigned kb = (unsigned)(k0 * ESZ);
        const int tap_off = ((dy * p.Wd + dx) * p.Cin + c0) * ESZ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned wv = sr.w_voff[j] == OOB_OFFSET ? OOB_OFFSET : sr.w_voff[j] + kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sr.w_rsrc, (lds_ptr_t)(wdst + j * 1024), 16, wv, 0, 0, 0);
            unsigned av;
            if (GATHER == G_LINEAR) {
                av = sr.a_voff[j] == OOB_OFFSET ? OOB_OFFSET : sr.a_voff[j] + kb;
            } else {
                const int yy = sr.a_y[j] + dy, xx = sr.a_x[j] + dx;
                const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
                av = ok ? (unsigned)((int)sr.a_voff[j] + tap_off) : OOB_OFFSET;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sr.a_rsrc, (lds_ptr_t)(adst + j * 1024), 16, av, 0, 0, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int z, m0, n0;
    decode_tile(tile, z, m0, n0);
    setup_stage(z, m0, n0);
    stage(0, 0);
    __syncthreads();  // the workgroup release waits for the DMA (vmcnt(0)) before the barrier
    int it = 0;       // global K-step counter: buffer parity runs across tile seams

    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int next_tile = tile + gridDim.x;
        int nz = 0, nm0 = 0, nn0 = 0;
        for (int kt = 0; kt < nk; ++kt, ++it) {
            const int buf = it & 1;
            // next K-tile's DMA flies under this tile's MFMAs -- across the seam it is the next OUTPUT tile's first
            {
                if (kt + 1 < nk) {
                    stage(kt + 1, buf ^ 1);
                } else if (next_tile < ntiles) {
                    decode_tile(next_tile, nz, nm0, nn0);
                    setup_stage(nz, nm0, nn0);
                    stage(0, buf ^ 1);
                }
            }
            const char* wbase = smem + buf * BUF_BYTES;
            const char* abase = wbase + TILE_BYTES;
            if constexpr (sizeof(T) == 4 && !__is_same(T, float)) {
                // split fp16: chunk f_kg = this lane's eight hi halves of the K = 32 block, chunk 4 + f_kg the lo halves
                u32x4 fwh[4], fwl[4], fah[4], fal[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    fwh[f] = *reinterpret_cast<const u32x4*>(wbase + swz(wy * 64 + f * 16 + f_row, f_kg));
                    fwl[f] = *reinterpret_cast<const u32x4*>(wbase + swz(wy * 64 + f * 16 + f_row, 4 + f_kg));
                    fah[f] = *reinterpret_cast<const u32x4*>(abase + swz(wx * 64 + f * 16 + f_row, f_kg));
                    fal[f] = *reinterpret_cast<const u32x4*>(abase + swz(wx * 64 + f * 16 + f_row, 4 + f_kg));
                }
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        acc[nf][mf] = split_mma(__builtin_bit_cast(f16x8, fwh[nf]), __builtin_bit_cast(f16x8, fwl[nf]),
                                                __builtin_bit_cast(f16x8, fah[mf]), __builtin_bit_cast(f16x8, fal[mf]), acc[nf][mf]);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {  // two 4-chunk groups per 128-byte row
                    u32x4 fw[4], fa[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        fw[f] = *reinterpret_cast<const u32x4*>(wbase + swz(wy * 64 + f * 16 + f_row, ks * 4 + f_kg));
                        fa[f] = *reinterpret_cast<const u32x4*>(abase + swz(wx * 64 + f * 16 + f_row, ks * 4 + f_kg));
                    }
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = mma(fw[nf], fa[mf], acc[nf][mf], T{});
                }
            }
            __syncthreads();
        }
        // the buffer consumed last, (it - 1) & 1, is free: the C tile is staged there; the other one already
        // holds the next output tile's first K-tile.
        char* cst = smem + ((it - 1) & 1) * BUF_BYTES;

        // ---- epilogue. Accumulator layout: lane holds n = nbase + 4*e_kg + (0..3) for m = mbase + e_row.
        // (the lane id is laundered through an empty asm: otherwise LICM hoists every epilogue address out of the
        // persistent tile loop, ~190 VGPRs stay live across the K-loop and the fragment reads serialise)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int e_row = lane_e & 15, e_kg = lane_e >> 4, tid_e = (tid & ~63) | lane_e;
        const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias_z : nullptr;
        char* __restrict__ Cb = reinterpret_cast<char*>(p.C);
        const size_t c_z = (size_t)z * p.strideC_z;
        // bias + activation are applied fragment by fragment right where a value is written out, so at most one
        // erff expansion is in flight and the accumulators die as they go (keeps the kernel at 2 workgroups / CU)
        f32x4 bvec[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = n0 + wy * 64 + nf * 16 + e_kg * 4;
            bvec[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) bvec[nf][j] = bias[n + j];
            }
        }
        auto finish = [&](f32x4 v, int nf) -> f32x4 {
            v = v * p.w_inv + bvec[nf];  // (w_inv: the power-of-two scale of split-fp16 Linear weights, 1 otherwise - exact)
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __is_same(T, SplitH) ? gelu_erfc_as(v[j]) : gelu_erf(v[j]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __is_same(T, SplitH) ? relu_keep_nan(v[j]) : fmaxf(v[j], 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
            return v;
        };

        if (p.planar_P > 0) {  // (B, N, P) fp32 planes from pixel-major rows (N is tiny: the 17 keypoint logits)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int m = m0 + wx * 64 + mf * 16 + e_row;
                if (m < p.M) {
                    const int img = m / p.planar_P, pix = m - img * p.planar_P;
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const int n = n0 + wy * 64 + nf * 16 + e_kg * 4;
                        float* o = reinterpret_cast<float*>(Cb) + c_z + ((size_t)img * p.N + n) * p.planar_P + pix;
                        const f32x4 v = finish(acc[nf][mf], nf);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (n + j < p.N) o[(size_t)j * p.planar_P] = v[j];
                    }
                }
            }
        } else {
            // Row-major output through LDS: 16-byte chunks of a row XOR-swizzled by (row & 15) -> the fragment
            // writes and the row reads are both bank-conflict free. bf16: whole 128 x 256 B tile at once;
            // fp32 / split fp16: two 64-row halves staged as fp32 (64 x 512 B = one 32 KiB buffer each); a split row
            // leaves as 8 elements per lane = 16 bytes of hi halves + 16 bytes of lo halves.
            constexpr bool OUT_BF16 = OUT == FMT_BF16, OUT_SPLIT = OUT == FMT_SPLIT;
            constexpr int halves = OUT_BF16 ? 1 : 2;
            constexpr int lpr = OUT == FMT_F32 ? 32 : 16;  // lanes per output row
            constexpr int epl = OUT == FMT_F32 ? 4 : 8;    // elements per lane
            constexpr int rows_per_pass = GEMM_THREADS / lpr;
            const int cl = tid_e % lpr, rl = tid_e / lpr;
            const int n = n0 + cl * epl;
#pragma unroll
            for (int h = 0; h < halves; ++h) {
                if (OUT_BF16 || wx == h) {
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf) {
                        const int ml = (OUT_BF16 ? wx * 64 : 0) + mf * 16 + e_row;  // row inside the staged block
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) {
                            const int nl = wy * 64 + nf * 16 + e_kg * 4;
                            const f32x4 v = finish(acc[nf][mf], nf);
                            if constexpr (OUT_BF16) {
                                const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                                const int byte = nl * 2;
                                *reinterpret_cast<bf16x4*>(cst + ml * 256 + ((((byte >> 4) ^ (ml & 15)) << 4) | (byte & 15))) = ov;
                            } else {
                                const int chunk = nl >> 2;
                                *reinterpret_cast<f32x4*>(cst + ml * 512 + ((chunk ^ (ml & 15)) << 4)) = v;
                            }
                        }
                    }
                }
                __syncthreads();
                constexpr int nrows = OUT_BF16 ? BM : BM / 2;
                for (int r0 = 0; r0 < nrows; r0 += rows_per_pass) {
                    const int ml = r0 + rl;
                    const int m = m0 + h * 64 + ml;
                    if (m >= p.M || n >= p.N) continue;
                    size_t orow = m;
                    if (GATHER == G_DECONV) {  // phase-interleaved output pixel (2y+py, 2x+px) of a (2H, 2W) map
                        const int hw = p.H * p.Wd;
                        const int b = m / hw, r = m - b * hw;
                        const int y = r / p.Wd, x = r - y * p.Wd;
                        const int py = p.py < 0 ? (z >> 1) : p.py, px = p.py < 0 ? (z & 1) : p.px;
                        orow = ((size_t)b * (2 * p.H) + 2 * y + py) * (2 * p.Wd) + 2 * x + px;
                    }
                    const size_t eoff = c_z + orow * p.ldc + n;
                    const bool full = n + epl <= p.N;
                    const size_t roff = p.res_mod > 0 ? (size_t)(m % p.res_mod) * p.ldres + n : c_z + orow * p.ldres + n;
                    if constexpr (OUT_BF16) {
                        u32x4 raw = *reinterpret_cast<const u32x4*>(cst + ml * 256 + ((cl ^ (ml & 15)) << 4));
                        __bf16* o = reinterpret_cast<__bf16*>(Cb) + eoff;
                        if (p.residual) {  // bf16 output with an fp32 residual: add in fp32, round once more
                            bf16x8 cv = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (n + j < p.N) cv[j] = (__bf16)((float)cv[j] + p.residual[roff + j]);
                            raw = __builtin_bit_cast(u32x4, cv);
                        }
                        if (full) {
                            *reinterpret_cast<u32x4*>(o) = raw;
                        } else {
                            const bf16x8 cv = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (n + j < p.N) o[j] = cv[j];
                        }
                    } else if constexpr (OUT_SPLIT) {
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(cst + ml * 512 + (((2 * cl) ^ (ml & 15)) << 4));
                        f32x4 v1 = *reinterpret_cast<const f32x4*>(cst + ml * 512 + (((2 * cl + 1) ^ (ml & 15)) << 4));
                        if (p.residual) {
                            v0 += *reinterpret_cast<const f32x4*>(p.residual + roff);
                            v1 += *reinterpret_cast<const f32x4*>(p.residual + roff + 4);
                        }
                        f16x8 hv, lv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hv[j] = split_hi(v0[j]);
                            lv[j] = split_lo(v0[j], hv[j]);
                            hv[4 + j] = split_hi(v1[j]);
                            lv[4 + j] = split_lo(v1[j], hv[4 + j]);
                        }
                        char* o = split_addr(Cb, eoff);  // N % 32 == 0 and ldc % 32 == 0 (checked at launch): always a full 8
                        *reinterpret_cast<f16x8*>(o) = hv;
                        *reinterpret_cast<f16x8*>(o + 64) = lv;
                    } else {
                        f32x4 v = *reinterpret_cast<const f32x4*>(cst + ml * 512 + ((cl ^ (ml & 15)) << 4));
                        float* o = reinterpret_cast<float*>(Cb) + eoff;
                        if (p.residual) {
                            if (full) {
                                v += *reinterpret_cast<const f32x4*>(p.residual + roff);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (n + j < p.N) v[j] += p.residual[roff + j];
                            }
                        }
                        if (full) {
                            *reinterpret_cast<f32x4*>(o) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (n + j < p.N) o[j] = v[j];
                        }
                    }
                }
                __syncthreads();  // the staging buffer is reused (next half / next tile's K-steps)
            }
        }
        z = nz;
        m0 = nm0;
        n0 = nn0;
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

template <typename T>
static int launch_gemm(const GemmParams& p_in, int groups, hipStream_t s) {
    GemmParams p = p_in;
    p.groups = groups;
    if (p.ksplit < 1) p.ksplit = 1;
    PP_REQUIRE(groups % p.ksplit == 0, PP_ERR_INVALID_ARG, "pp gemm: groups must be a multiple of ksplit");
    constexpr int BK = Prec<T>::BK;
    PP_REQUIRE(p.K > 0 && p.K % BK == 0, PP_ERR_UNSUPPORTED, "pp gemm: K must be a positive multiple of the K-tile");
    PP_REQUIRE(p.M > 0 && p.N > 0, PP_ERR_INVALID_ARG, "pp gemm: M and N must be positive");
    PP_REQUIRE(p.out_bf16 >= FMT_F32 && p.out_bf16 <= FMT_SPLIT, PP_ERR_INVALID_ARG, "pp gemm: unknown output format");
    constexpr bool is_split = !__is_same(T, float) && sizeof(T) == 4;
    PP_REQUIRE(p.out_bf16 == FMT_F32 || (p.out_bf16 == FMT_SPLIT) == is_split, PP_ERR_UNSUPPORTED,
               "pp gemm: the output is fp32 or the operand format of the precision mode (bf16 / split fp16)");
    PP_REQUIRE(p.planar_P > 0 || p.ldc % (p.out_bf16 ? 8 : 4) == 0, PP_ERR_UNSUPPORTED,
               "pp gemm: ldc must be a multiple of 8 (bf16 out) / 4 (fp32 out) elements");
    PP_REQUIRE(p.out_bf16 != FMT_SPLIT || (p.ldc % 32 == 0 && p.N % 32 == 0 && p.strideC_z % 32 == 0), PP_ERR_UNSUPPORTED,
               "pp gemm: split-fp16 output needs N, ldc and the group stride to be multiples of 32 elements");
    if (is_split)
        PP_REQUIRE(p.lda % 32 == 0 && p.ldw % 32 == 0 && p.strideA_z % 32 == 0 && p.strideW_z % 32 == 0, PP_ERR_UNSUPPORTED,
                   "pp gemm: split-fp16 operands need row pitches and group strides that are multiples of 32 elements");
    PP_REQUIRE(!(p.planar_P > 0 && p.out_bf16), PP_ERR_UNSUPPORTED, "pp gemm: planar output is fp32 only");
    if (p.gather != G_LINEAR)
        PP_REQUIRE(p.Cin % BK == 0 && p.H > 0 && p.Wd > 0, PP_ERR_UNSUPPORTED,
                   "pp gemm: conv gathers need Cin to be a multiple of the K-tile");
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < OOB_OFFSET && p.w_bytes < OOB_OFFSET, PP_ERR_UNSUPPORTED,
               "pp gemm: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const long long ntiles = (long long)((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * groups;
    PP_REQUIRE(ntiles < (1ll << 30), PP_ERR_UNSUPPORTED, "pp gemm: too many output tiles");
    const size_t lds = 2 * BUF_BYTES;  // 64 KiB: two workgroups per CU
    int slots = 2 * device_cus();      // persistent grid; a multiple of 8 keeps a workgroup's tiles on one XCD
    slots -= slots % 8;
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    void (*kern)(const GemmParams) = nullptr;
    const bool ob = p.out_bf16 != 0;
    constexpr int OP = __is_same(T, float) ? FMT_F32 : (is_split ? FMT_SPLIT : FMT_BF16);  // operand-format output of T
    switch (p.gather) {
        case G_LINEAR: kern = ob ? gemm_kernel<T, G_LINEAR, OP> : gemm_kernel<T, G_LINEAR, FMT_F32>; break;
        case G_CONV3: kern = ob ? gemm_kernel<T, G_CONV3, OP> : gemm_kernel<T, G_CONV3, FMT_F32>; break;
        case G_DECONV: kern = ob ? gemm_kernel<T, G_DECONV, OP> : gemm_kernel<T, G_DECONV, FMT_F32>; break;
        default: return fail(PP_ERR_INVALID_ARG, "pp gemm: unknown gather mode");
    }
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GEMM_THREADS), lds, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int gemm(const GemmParams& p, int prec, int groups, hipStream_t s) {
    if (prec == PP_PREC_BF16) return launch_gemm<__bf16>(p, groups, s);
    if (prec == PP_PREC_F32) return launch_gemm<float>(p, groups, s);
    if (prec == PP_PREC_F16X3) return launch_gemm<SplitH>(p, groups, s);
    return fail(PP_ERR_INVALID_ARG, "pp gemm: unknown precision");
}

}  // namespace pp

static bool panel_enabled() {
    const bool use_panel = pp::option("panel") != 0;  // dev switch for A/B timing (pp_set_option)
    return use_panel;
}

// ---------------------------------------------------------------------------------------------
extern "C" int pp_gemm(int prec, const void* act, const void* weight, const float* bias, const float* residual,
                       int res_mod, void* out, int M, int N, int K, int lda, int ldw, int ldc, int act_fn,
                       int out_bf16, int planar_P, void* stream) {
    return pp_gemm_ws(prec, act, weight, bias, residual, res_mod, out, M, N, K, lda, ldw, ldc, act_fn, out_bf16, planar_P, 1.0f, stream);
}

extern "C" int pp_gemm_ws(int prec, const void* act, const void* weight, const float* bias, const float* residual,
                          int res_mod, void* out, int M, int N, int K, int lda, int ldw, int ldc, int act_fn,
                          int out_bf16, int planar_P, float w_inv_scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(act && weight && out, PP_ERR_INVALID_ARG, "pp_gemm: act, weight and out must be non-NULL");
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_gemm: the weight scale must be a power of two in [2^-40, 2^40]");
        PP_REQUIRE(w_inv_scale == 1.0f || prec == PP_PREC_F16X3, PP_ERR_INVALID_ARG, "pp_gemm: weight scales belong to the split-fp16 mode");
    }
    GemmParams p{};
    p.w_inv = w_inv_scale;
    p.A = act; p.W = weight; p.C = out; p.bias = bias; p.residual = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.act = act_fn; p.out_bf16 = out_bf16; p.gather = G_LINEAR;
    p.res_mod = res_mod; p.ldres = ldc; p.planar_P = planar_P;
    const size_t esz = prec == PP_PREC_BF16 ? 2 : 4;
    PP_REQUIRE(M > 0 && N > 0 && K > 0, PP_ERR_INVALID_ARG, "pp_gemm: M, N and K must be positive");
    const size_t ab = ((size_t)(M - 1) * lda + K) * esz, wb = ((size_t)(N - 1) * ldw + K) * esz;
    PP_REQUIRE(ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_gemm: operands must be smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    PP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, PP_ERR_UNSUPPORTED, "pp_gemm: lda/ldw must be multiples of 8 elements");
    if (panel_enabled() && linear_dma_supported(p, prec, 1)) return linear_dma_gemm(p, reinterpret_cast<hipStream_t>(stream));
    if (panel_enabled() && panel_split_supported(p, prec, 1)) return panel_split_gemm(p, prec, 1, reinterpret_cast<hipStream_t>(stream));
    return gemm(p, prec, 1, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_conv_gemm(int prec, int kind, const void* act_nhwc, const void* weight, const float* bias, void* out,
                            int B, int H, int W, int Cin, int Cout, int py, int px, int groups,
                            long long stride_act_g, long long stride_w_g, long long stride_out_g,
                            long long stride_bias_g, int ldc, int act_fn, int out_bf16, void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && out, PP_ERR_INVALID_ARG, "pp_conv_gemm: act, weight and out must be non-NULL");
    PP_REQUIRE(kind == PP_CONV3X3 || kind == PP_DECONV4X4S2, PP_ERR_INVALID_ARG, "pp_conv_gemm: unknown kind");
    PP_REQUIRE(groups >= 1 && B > 0, PP_ERR_INVALID_ARG, "pp_conv_gemm: bad groups/B");
    if (kind == PP_DECONV4X4S2 && py < 0) {  // all four output phases in one launch: group g = phase (g >> 1, g & 1)
        PP_REQUIRE(groups == 1, PP_ERR_INVALID_ARG, "pp_conv_gemm: py < 0 (all deconv phases) excludes explicit groups");
        groups = 4;
        stride_act_g = 0;
        stride_out_g = 0;
        stride_bias_g = 0;
        stride_w_g = (long long)Cout * 4 * Cin;
    }
    PP_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cout > 0, PP_ERR_INVALID_ARG, "pp_conv_gemm: bad shape");
    GemmParams p{};
    p.A = act_nhwc; p.W = weight; p.C = out; p.bias = bias; p.residual = nullptr;
    p.M = B * H * W; p.N = Cout;
    p.K = (kind == PP_CONV3X3 ? 9 : 4) * Cin;
    p.lda = Cin; p.ldw = p.K; p.ldc = ldc;
    p.H = H; p.Wd = W; p.Cin = Cin; p.py = py; p.px = px;
    p.act = act_fn; p.out_bf16 = out_bf16;
    p.gather = kind == PP_CONV3X3 ? G_CONV3 : G_DECONV;
    p.ldres = ldc;
    const size_t esz = prec == PP_PREC_BF16 ? 2 : 4;
    const size_t ab = (size_t)B * H * W * Cin * esz, wb = (size_t)Cout * p.K * esz;
    PP_REQUIRE(ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_conv_gemm: operands must be smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.strideA_z = stride_act_g; p.strideW_z = stride_w_g; p.strideC_z = stride_out_g; p.strideBias_z = stride_bias_g;
    if (panel_enabled() && panel_split_supported(p, prec, groups)) return panel_split_gemm(p, prec, groups, reinterpret_cast<hipStream_t>(stream));
    if (panel_enabled() && conv_halo_supported(p, prec, groups)) return conv_halo(p, groups, reinterpret_cast<hipStream_t>(stream));
    if (panel_enabled() && panel_gemm_supported(p, prec, groups)) return panel_gemm(p, groups, reinterpret_cast<hipStream_t>(stream));
    return gemm(p, prec, groups, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_conv3x3_splitk(int prec, const void* act_nhwc, const void* weight, float* partials, int B, int H, int W,
                                 int Cin, int Cout, int groups, long long stride_act_g, long long stride_w_g, int ksplit,
                                 void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && partials, PP_ERR_INVALID_ARG, "pp_conv3x3_splitk: NULL argument");
    PP_REQUIRE(groups >= 1 && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PP_ERR_INVALID_ARG, "pp_conv3x3_splitk: bad shape");
    PP_REQUIRE(ksplit >= 1 && ksplit <= 16 && (ksplit == 1 || ksplit == 3 || ksplit == 9 || Cin % (32 * ksplit) == 0), PP_ERR_UNSUPPORTED,
               "pp_conv3x3_splitk: ksplit must be 1, 3 or 9 (whole taps), or a count of channel ranges (Cin % (32 ksplit) == 0, split-fp16 wide-tile kernel)");
    GemmParams p{};
    p.A = act_nhwc; p.W = weight; p.C = partials; p.bias = nullptr; p.residual = nullptr;
    p.M = B * H * W; p.N = Cout;
    p.K = 9 * Cin / ksplit;
    p.lda = Cin; p.ldw = 9 * Cin; p.ldc = Cout;
    p.H = H; p.Wd = W; p.Cin = Cin;
    p.act = ACT_NONE; p.out_bf16 = 0; p.gather = G_CONV3; p.ldres = Cout;
    p.ksplit = ksplit;
    const size_t esz = prec == PP_PREC_BF16 ? 2 : 4;
    const size_t ab = (size_t)B * H * W * Cin * esz, wb = (size_t)Cout * 9 * Cin * esz;
    PP_REQUIRE(ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_conv3x3_splitk: operands must be smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.strideA_z = stride_act_g; p.strideW_z = stride_w_g; p.strideC_z = (long long)p.M * Cout; p.strideBias_z = 0;
    // Channel-range slices (pp_conv3x3_splitk_slices picks them for the split-fp16 4 x 4 tower stage: four quarters of the channels on
    // 256 x 192 tiles = one workgroup per CU): the wide-tile split kernel only. Whole-tap slices (3, 9): the kernels below.
    if (ksplit != 1 && ksplit != 3 && ksplit != 9) {
        PP_REQUIRE(panel_enabled() && panel_split_supported(p, prec, groups * ksplit), PP_ERR_UNSUPPORTED,
                   "pp_conv3x3_splitk: this slice count is a channel-range split, built for the split-fp16 wide-tile kernel only (use pp_conv3x3_splitk_slices)");
        return panel_split_gemm(p, prec, groups * ksplit, reinterpret_cast<hipStream_t>(stream));
    }
    if (panel_enabled() && panel_gemm_supported(p, prec, groups * ksplit))  // enough 256 x 192 tiles to fill the chip
        return panel_gemm(p, groups * ksplit, reinterpret_cast<hipStream_t>(stream));
    return gemm(p, prec, groups * ksplit, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_deconv_head(const void* act_nhwc, const void* weight, const float* bias, const void* head_w,
                              const float* head_b, float* logits_phased, int B, int H, int W, int Cin, int Cout, int K,
                              void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && head_w && head_b && logits_phased, PP_ERR_INVALID_ARG, "pp_deconv_head: NULL argument");
    PP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0, PP_ERR_INVALID_ARG, "pp_deconv_head: bad shape");
    PP_REQUIRE(Cout == 256 && K >= 1 && K <= 28 && Cin % 32 == 0 && (4 * Cin) % 128 == 0 && (H * W) % 4 == 0, PP_ERR_UNSUPPORTED,
               "pp_deconv_head: built for 256 deconvolution channels, at most 28 output maps, Cin % 32 == 0, H*W % 4 == 0");
    GemmParams p{};
    p.A = act_nhwc; p.W = weight; p.C = logits_phased; p.bias = bias; p.residual = nullptr;
    p.M = B * H * W; p.N = Cout; p.K = 4 * Cin;
    p.lda = Cin; p.ldw = p.K; p.ldc = Cout;
    p.H = H; p.Wd = W; p.Cin = Cin; p.py = -1; p.px = -1;
    p.act = ACT_RELU; p.out_bf16 = 1; p.gather = G_DECONV; p.ldres = Cout;
    p.head_w = head_w; p.head_b = head_b; p.head_out = logits_phased; p.head_n = K;
    const size_t ab = (size_t)B * H * W * Cin * 2, wb = (size_t)Cout * p.K * 2;
    PP_REQUIRE(ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_deconv_head: operands must be smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.strideA_z = 0; p.strideW_z = (long long)Cout * 4 * Cin; p.strideC_z = 0; p.strideBias_z = 0;
    return panel_gemm(p, 4, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_deconv_head_split(const void* act_nhwc, const void* weight, const float* bias, const void* head_w_packed,
                                    const float* head_b, float* logits_phased, int B, int H, int W, int Cin, int Cout, int K,
                                    void* stream) {
    using namespace pp;
    PP_REQUIRE(act_nhwc && weight && head_w_packed && head_b && logits_phased, PP_ERR_INVALID_ARG, "pp_deconv_head_split: NULL argument");
    PP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0, PP_ERR_INVALID_ARG, "pp_deconv_head_split: bad shape");
    PP_REQUIRE(Cout == 256 && K >= 1 && K <= 28 && Cin % 32 == 0, PP_ERR_UNSUPPORTED,
               "pp_deconv_head_split: built for 256 deconvolution channels, at most 28 output maps, Cin % 32 == 0");
    GemmParams p{};
    p.A = act_nhwc; p.W = weight; p.C = logits_phased; p.bias = bias; p.residual = nullptr;
    p.M = B * H * W; p.N = Cout; p.K = 4 * Cin;
    p.lda = Cin; p.ldw = p.K; p.ldc = Cout;
    p.H = H; p.Wd = W; p.Cin = Cin; p.py = -1; p.px = -1;
    p.act = ACT_RELU; p.out_bf16 = 2; p.gather = G_DECONV; p.ldres = Cout;
    p.head_w = head_w_packed; p.head_b = head_b; p.head_out = logits_phased; p.head_n = K;
    const size_t ab = (size_t)B * H * W * Cin * 4, wb = (size_t)Cout * p.K * 4;
    PP_REQUIRE(ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_deconv_head_split: operands must be smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.strideA_z = 0; p.strideW_z = (long long)Cout * 4 * Cin; p.strideC_z = 0; p.strideBias_z = 0;
    PP_REQUIRE(panel_split_supported(p, PP_PREC_F16X3, 4), PP_ERR_UNSUPPORTED,
               "pp_deconv_head_split: needs at least 192 tiles of 192 pixels x 256 channels over the four phases");
    return panel_split_gemm(p, PP_PREC_F16X3, 4, reinterpret_cast<hipStream_t>(stream));
}
