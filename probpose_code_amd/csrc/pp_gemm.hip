// MFMA GEMM family for gfx950: every dense contraction of the path (ViT linears, patch embed,
// deconv phases, 3x3 tower convs, 1x1 convs) is  C[m, n] = sum_k Act[m, k] * Wt[n, k]  with
// a K-contiguous weight matrix Wt (the nn.Linear layout) and an activation operand that is
// either a plain row-major matrix or an implicit im2col view of an NHWC tensor (no im2col
// buffer is ever materialised for the convolutions).
//
//   * 128x128 block tile, 256 threads = 4 waves as 2(n) x 2(m), each wave 64x64 = 4x4 MFMA
//     16x16 fragments, fp32 accumulators in registers;
//   * operand precision is a template parameter:
//       bf16  -> v_mfma_f32_16x16x32_bf16, BK = 64 (128-byte LDS rows)
//       f32   -> v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), BK = 32
//     both have the same 128-byte-per-row LDS image and the same C/D fragment layout, so the
//     staging, swizzle and epilogue code is shared;
//   * global -> registers -> LDS staging, issued one K-tile ahead of the MFMAs (loads fly
//     under the matrix work; writes land after it), 2 LDS buffers, one barrier per K-tile;
//   * LDS rows are 8 x 16-byte chunks, chunk index XOR-ed with (row & 7): the ds_read_b128
//     fragment reads and the ds_write_b128 staging writes are both bank-conflict free;
//   * the weight tile is the MFMA "A" operand and the activation tile the "B" operand, so a
//     lane ends up with 4 consecutive n for one m: bias/residual/output move as 16-byte
//     (fp32) or 8-byte (bf16) vectors;
//   * fused epilogue: + bias[n], exact-erf GELU or ReLU, + fp32 residual, fp32 or bf16 store,
//     optional output-row remap (deconv phase interleave).
#include "pp_common.h"
#include "pp_gemm.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int GEMM_THREADS = 256;
constexpr int ROW_BYTES = 128;                  // one K-tile row in LDS
constexpr int TILE_BYTES = BM * ROW_BYTES;      // 16 KiB per operand per buffer

template <typename T>
struct Prec;
template <>
struct Prec<__bf16> {
    static constexpr int BK = 64;   // elements per K-tile
    static constexpr int CH = 8;    // elements per 16-byte chunk
};
template <>
struct Prec<float> {
    static constexpr int BK = 32;
    static constexpr int CH = 4;
};

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

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned OOB_OFFSET = 0x7ffffff0u;  // >= num_records of every tensor (< 2 GiB): the DMA writes zeros

template <typename T, int GATHER>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(const GemmParams p) {
    constexpr int BK = Prec<T>::BK;
    constexpr int ESZ = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2 buffers][W tile | Act tile], each tile 128 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int z = blockIdx.z;

    // XCD-aware tile order: hardware deals consecutive block ids round-robin to the 8 XCDs; give each XCD a
    // contiguous run of tiles so that the n-tiles sharing one activation panel hit the same L2.
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, ntiles = ntn * ntm;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int n0 = (tile % ntn) * BN;
    const int m0 = (tile / ntn) * BM;

    const char* Act = reinterpret_cast<const char*>(p.A) + (size_t)z * p.strideA_z * ESZ;
    const char* Wt = reinterpret_cast<const char*>(p.W) + (size_t)z * p.strideW_z * ESZ;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Act), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, p.w_bytes, 0x00020000);

    // ---- LDS-DMA staging: one `buffer_load_dwordx4 ... lds` moves 64 lanes x 16 B = 8 tile rows; wave w
    // issues rows [32 w + 8 j, +8), j = 0..3, of both tiles. LDS destination is lane-linear, so the swizzle
    // goes on the SOURCE: lane i lands in row 8j' + (i >> 3), slot (i & 7) and must fetch chunk (i & 7) ^ (i >> 3).
    const int d_row = lane >> 3;
    const unsigned d_chunk_bytes = (unsigned)(((lane & 7) ^ d_row) << 4);
    unsigned a_voff[4], w_voff[4];
    int a_y[4], a_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 32 + j * 8 + d_row;
        const int n = n0 + r, m = m0 + r;
        w_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.ldw * ESZ) + d_chunk_bytes : OOB_OFFSET;
        a_y[j] = a_x[j] = 0;
        if (GATHER == G_LINEAR) {
            a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.lda * ESZ) + d_chunk_bytes : OOB_OFFSET;
        } else {
            const int hw = p.H * p.Wd;
            const int b = m / hw, rr = m - b * hw;
            a_y[j] = m < p.M ? rr / p.Wd : -100000;  // tail rows fail every bounds test below
            a_x[j] = rr - (rr / p.Wd) * p.Wd;
            a_voff[j] = (unsigned)m * (unsigned)(p.Cin * ESZ) + d_chunk_bytes;  // NHWC pixel origin
        }
    }

    auto stage = [&](int kt, int buf) {
        const int k0 = kt * BK;
        char* wdst = smem + buf * 2 * TILE_BYTES + wave * 4096;
        char* adst = wdst + TILE_BYTES;
        int dy = 0, dx = 0, c0 = k0;
        if (GATHER != G_LINEAR) {
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin;
            if (GATHER == G_CONV3) {  // 3x3, pad 1: tap = ky*3 + kx reads (y + ky - 1, x + kx - 1)
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            } else {  // deconv k4 s2 p1, output phase (py, px): tap = ty*2 + tx reads (y + ty - 1 + py, x + tx - 1 + px)
                dy = (tap >> 1) - 1 + p.py;
                dx = (tap & 1) - 1 + p.px;
            }
        }
        const unsigned kb = (unsigned)(k0 * ESZ);
        const int tap_off = ((dy * p.Wd + dx) * p.Cin + c0) * ESZ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned wv = w_voff[j] == OOB_OFFSET ? OOB_OFFSET : w_voff[j] + kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(wdst + j * 1024), 16, wv, 0, 0, 0);
            unsigned av;
            if (GATHER == G_LINEAR) {
                av = a_voff[j] == OOB_OFFSET ? OOB_OFFSET : a_voff[j] + kb;
            } else {
                const int yy = a_y[j] + dy, xx = a_x[j] + dx;
                const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
                av = ok ? (unsigned)((int)a_voff[j] + tap_off) : OOB_OFFSET;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(adst + j * 1024), 16, av, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();  // the workgroup release waits for the DMA (vmcnt(0)) before the barrier
    const int f_row = lane & 15, f_kg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);  // next tile's DMA flies under this tile's MFMAs
        const char* wbase = smem + buf * 2 * TILE_BYTES;
        const char* abase = wbase + TILE_BYTES;
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
        __syncthreads();
    }

    // ---- epilogue. Accumulator layout: lane holds n = nbase + 4*f_kg + (0..3) for m = mbase + f_row.
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias_z : nullptr;
    char* __restrict__ Cb = reinterpret_cast<char*>(p.C);
    const size_t c_z = (size_t)z * p.strideC_z;
    // bias + activation in registers (bias depends on n only: 4 loads per lane, not 16)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        const int n = n0 + wy * 64 + nf * 16 + f_kg * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) bv[j] = bias[n + j];
        }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            f32x4 v = acc[nf][mf] + bv;
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            acc[nf][mf] = v;
        }
    }

    if (p.planar_P > 0) {  // (B, N, P) fp32 planes from pixel-major rows (N is tiny: the 17 keypoint logits)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int m = m0 + wx * 64 + mf * 16 + f_row;
            if (m >= p.M) continue;
            const int img = m / p.planar_P, pix = m - img * p.planar_P;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int n = n0 + wy * 64 + nf * 16 + f_kg * 4;
                float* o = reinterpret_cast<float*>(Cb) + c_z + ((size_t)img * p.N + n) * p.planar_P + pix;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) o[(size_t)j * p.planar_P] = acc[nf][mf][j];
            }
        }
        return;
    }

    // Row-major output: stage the 128x128 tile through LDS (the staging buffers are free now) so that every
    // output row leaves the CU as one contiguous 256-B (bf16) / 512-B (fp32) run of 16-byte lane stores,
    // instead of 32-byte scraps per MFMA fragment. Rows are padded by 16 B: conflict-free b64/b128 writes.
    const int osz = p.out_bf16 ? 2 : 4;
    const int crow = BN * osz + 16;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int ml = wx * 64 + mf * 16 + f_row;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int nl = wy * 64 + nf * 16 + f_kg * 4;
            const f32x4 v = acc[nf][mf];
            if (p.out_bf16) {
                const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4*>(smem + ml * crow + nl * 2) = ov;
            } else {
                *reinterpret_cast<f32x4*>(smem + ml * crow + nl * 4) = v;
            }
        }
    }
    __syncthreads();
    {
        const int lpr = BN * osz / 16;            // lanes per output row: 16 (bf16) or 32 (fp32)
        const int epl = 16 / osz;                 // elements per lane: 8 or 4
        const int rows_per_pass = GEMM_THREADS / lpr;
        const int cl = tid % lpr, rl = tid / lpr;
        const int n = n0 + cl * epl;
        for (int r0 = 0; r0 < BM; r0 += rows_per_pass) {
            const int ml = r0 + rl;
            const int m = m0 + ml;
            if (m >= p.M || n >= p.N) continue;
            size_t orow = m;
            if (GATHER == G_DECONV) {  // phase-interleaved output pixel (2y+py, 2x+px) of a (2H, 2W) map
                const int hw = p.H * p.Wd;
                const int b = m / hw, r = m - b * hw;
                const int y = r / p.Wd, x = r - y * p.Wd;
                orow = ((size_t)b * (2 * p.H) + 2 * y + p.py) * (2 * p.Wd) + 2 * x + p.px;
            }
            const size_t eoff = c_z + orow * p.ldc + n;
            const bool full = n + epl <= p.N;
            if (p.out_bf16) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(smem + ml * crow + cl * 16);
                __bf16* o = reinterpret_cast<__bf16*>(Cb) + eoff;
                if (p.residual) {  // bf16 output with an fp32 residual: add in fp32, round once
                    bf16x8 cv = __builtin_bit_cast(bf16x8, raw);
                    const size_t roff = p.res_mod > 0 ? (size_t)(m % p.res_mod) * p.ldres + n : c_z + orow * p.ldres + n;
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
            } else {
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + ml * crow + cl * 16);
                float* o = reinterpret_cast<float*>(Cb) + eoff;
                if (p.residual) {
                    const size_t roff = p.res_mod > 0 ? (size_t)(m % p.res_mod) * p.ldres + n : c_z + orow * p.ldres + n;
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
    }
}

template <typename T>
static int launch_gemm(const GemmParams& p, int groups, hipStream_t s) {
    constexpr int BK = Prec<T>::BK;
    PP_REQUIRE(p.K > 0 && p.K % BK == 0, PP_ERR_UNSUPPORTED, "pp gemm: K must be a positive multiple of the K-tile");
    PP_REQUIRE(p.M > 0 && p.N > 0, PP_ERR_INVALID_ARG, "pp gemm: M and N must be positive");
    PP_REQUIRE(p.planar_P > 0 || p.ldc % (p.out_bf16 ? 8 : 4) == 0, PP_ERR_UNSUPPORTED,
               "pp gemm: ldc must be a multiple of 8 (bf16 out) / 4 (fp32 out) elements");
    PP_REQUIRE(!(p.planar_P > 0 && p.out_bf16), PP_ERR_UNSUPPORTED, "pp gemm: planar output is fp32 only");
    if (p.gather != G_LINEAR)
        PP_REQUIRE(p.Cin % BK == 0 && p.H > 0 && p.Wd > 0, PP_ERR_UNSUPPORTED,
                   "pp gemm: conv gathers need Cin to be a multiple of the K-tile");
    const dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, groups);
    PP_REQUIRE(p.a_bytes > 0 && p.w_bytes > 0 && p.a_bytes < OOB_OFFSET && p.w_bytes < OOB_OFFSET, PP_ERR_UNSUPPORTED,
               "pp gemm: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const size_t lds = 4 * TILE_BYTES + 2048;  // 2 buffers x (W tile + Act tile); the fp32 C tile needs 128 x 528 B
    void (*kern)(const GemmParams) = nullptr;
    switch (p.gather) {
        case G_LINEAR: kern = gemm_kernel<T, G_LINEAR>; break;
        case G_CONV3: kern = gemm_kernel<T, G_CONV3>; break;
        case G_DECONV: kern = gemm_kernel<T, G_DECONV>; break;
        default: return fail(PP_ERR_INVALID_ARG, "pp gemm: unknown gather mode");
    }
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int gemm(const GemmParams& p, int prec, int groups, hipStream_t s) {
    if (prec == PP_PREC_BF16) return launch_gemm<__bf16>(p, groups, s);
    if (prec == PP_PREC_F32) return launch_gemm<float>(p, groups, s);
    return fail(PP_ERR_INVALID_ARG, "pp gemm: unknown precision");
}

}  // namespace pp

// ---------------------------------------------------------------------------------------------
extern "C" int pp_gemm(int prec, const void* act, const void* weight, const float* bias, const float* residual,
                       int res_mod, void* out, int M, int N, int K, int lda, int ldw, int ldc, int act_fn,
                       int out_bf16, int planar_P, void* stream) {
    using namespace pp;
    PP_REQUIRE(act && weight && out, PP_ERR_INVALID_ARG, "pp_gemm: act, weight and out must be non-NULL");
    GemmParams p{};
    p.A = act; p.W = weight; p.C = out; p.bias = bias; p.residual = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.act = act_fn; p.out_bf16 = out_bf16; p.gather = G_LINEAR;
    p.res_mod = res_mod; p.ldres = ldc; p.planar_P = planar_P;
    const size_t esz = prec == PP_PREC_BF16 ? 2 : 4;
    const size_t ab = ((size_t)(M - 1) * lda + K) * esz, wb = ((size_t)(N - 1) * ldw + K) * esz;
    PP_REQUIRE(M > 0 && N > 0 && K > 0 && ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_gemm: operands must be non-empty and smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    PP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, PP_ERR_UNSUPPORTED, "pp_gemm: lda/ldw must be multiples of 8 elements");
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
    PP_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cout > 0 && ab < 0x7ffffff0u && wb < 0x7ffffff0u, PP_ERR_UNSUPPORTED,
               "pp_conv_gemm: operands must be non-empty and smaller than 2 GiB");
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.strideA_z = stride_act_g; p.strideW_z = stride_w_g; p.strideC_z = stride_out_g; p.strideBias_z = stride_bias_g;
    return gemm(p, prec, groups, reinterpret_cast<hipStream_t>(stream));
}
