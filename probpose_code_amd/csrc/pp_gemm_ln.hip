// Fused  x <- x + act @ W^T + bias ;  h <- LayerNorm(x)  for the ViT residual stream on gfx950.
//
// In the backbone every residual GEMM (attention proj, FFN fc2, patch embed) is followed by a LayerNorm over the
// E = 384 features of each token (mmpretrain VisionTransformer [3P]: x = x + attn(ln1(x)); x = ffn(ln2(x)) + x; the
// last layer feeds the final ln1). These GEMMs are memory-bound (N = E is only 384: arithmetic intensity far below the
// MFMA/HBM ridge), and a separate LayerNorm launch re-reads the fp32 stream it has just written. Here one workgroup owns
// 96 complete rows, so the LayerNorm statistics are available in registers right after the K-loop:
//
//   * grid = M / 96 workgroups of 768 threads (12 waves, 3 per SIMD); at bs 64 with flip test M = 24 576 -> 256
//     workgroups = one per CU, a single balanced round;
//   * wave (rw, cw) owns rows 32 rw .. +31 x columns 96 cw .. +95: 2 x 6 MFMA 16x16 fragments, fp32 accumulators
//     (round 1 had 16 x 192 wave tiles: 13 LDS fragment reads per 12 MFMAs - with twelve waves the LDS read port, 128 B/clk,
//     was busier than the matrix pipe; 32 x 96 reads 8 per 12);
//   * the activation K-tile (96 x 128 B) and the WHOLE weight K-tile (384 x 128 B) are staged by LDS-DMA into a
//     2-deep ring (120 KiB), chunk-swizzled on the source side exactly as in pp_gemm.hip; each activation byte is
//     read from HBM once, the weights stream from L2;
//   * epilogue in registers: + bias + residual (fp32, optionally a broadcast table = pos_embed), row mean and
//     centred variance by two 2-hop DPP reductions + one LDS exchange between the two column halves, then the fp32
//     stream and the normalised operand (bf16 or fp32) are both written out.
//
// E = 768 (ViT-B, BASELINE config 4; round 2): the same kernel with the 768 output columns as TWO halves of 384 that run
// through the K-loop one after the other (the activation tile is streamed a second time - it comes out of L2 - the weight
// tile of a stage stays 384 rows x 128 B, so the LDS image and the DMA pattern do not change), both halves' accumulators
// live in registers (112 rows x 768 columns over eight waves = 168 fp32 registers per lane) and the LayerNorm statistics
// run over both at the end. 112-row tiles: M = 27 648 rows at bs 32 with flip test = 247 workgroups, one round on 256 CUs
// (96-row tiles would be 288 = a second round for 32 workgroups). Replaces GEMM + residual (128 x 128 tiles) followed by a
// LayerNorm launch that re-reads the fp32 stream.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace rl {

constexpr int ROW_BYTES = 128;
constexpr int BNH = 384;                           // output columns per pass of the K-loop (one "half" at E = 768)
// BM rows per workgroup, NH column halves of 384, waves as RW row groups x CW column groups
template <int BM_, int NH_, int RW_, int CW_, int MINW_>
struct Cfg {
    static constexpr int BM = BM_, NH = NH_, BN = BNH * NH_, RW = RW_, CW = CW_, MINW = MINW_;
    static constexpr int WAVES = RW * CW, THREADS = 64 * WAVES;
    static constexpr int W_TILE = BNH * ROW_BYTES;            // 48 KiB
    static constexpr int A_TILE = BM * ROW_BYTES;             // 12 / 14 KiB
    static constexpr int STAGE = W_TILE + A_TILE;
    static constexpr int DMA_PER_STAGE = (BNH + BM) / 8;      // instructions of 8 rows
    static constexpr int DPW = (DMA_PER_STAGE + WAVES - 1) / WAVES;
    static constexpr int RFW = BM / 16 / RW, NFW = BNH / 16 / CW;  // wave tile per half: RFW x NFW fragments
    static_assert(RFW * RW * 16 == BM && NFW * CW * 16 == BNH, "wave tiling");
    static constexpr int STAT_BYTES = CW * BM * 4;            // row statistics exchanged between the column groups
    static constexpr int LDS = 2 * STAGE + 2 * STAT_BYTES;
};
typedef Cfg<96, 1, 3, 4, 3> C384;    // E = 384: 96 x 384, wave tile 32 x 96 (2 x 6 fragments)
typedef Cfg<112, 2, 1, 8, 2> C768;   // E = 768: 112 x (2 x 384), eight waves, wave tile 112 x 48 per half (7 x 3 fragments x 2 = 168 accumulator registers;
                                     // twelve waves = three per SIMD leave 168 registers per lane in all: 77 spilled)
constexpr unsigned OOB_OFFSET = 0x7ffffff0u;

struct Params {
    const void* A;         // [M, lda] activations (bf16 / fp32)
    const void* W;         // [384, ldw] weights, K contiguous
    const float* bias;     // [384] or NULL
    const float* residual; // fp32 [M, 384] (may alias x_out) or a [res_mod, 384] table
    float* x_out;          // fp32 [M, 384]
    void* h_out;           // LayerNorm(x_out), bf16 or fp32 [M, 384]
    const float* gamma;
    const float* beta;
    int M, K, lda, ldw, res_mod, h_bf16;
    unsigned a_bytes, w_bytes;
    float eps;
    float w_s = 1.0f, w_inv = 1.0f;  // split-fp16 weights stored as w * w_s (a power of two): residual + bias enter the accumulators times w_s, the sums leave times w_inv
};

template <typename T>
struct Prec;
template <>
struct Prec<__bf16> {
    static constexpr int BK = 64;
};
template <>
struct Prec<float> {
    static constexpr int BK = 32;
};
template <>
struct Prec<SplitH> {
    static constexpr int BK = 32;  // split fp16 (pp_split.h): one 128-byte block = 32 hi halves | 32 lo halves
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

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

template <typename T, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_res_ln_kernel(const Params p) {
    constexpr int BK = Prec<T>::BK;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BM = C::BM, BN = C::BN, NH = C::NH, RW = C::RW, CW = C::CW, RFW = C::RFW, NFW = C::NFW, DPW = C::DPW;
    constexpr int STAGE = C::STAGE, W_TILE = C::W_TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][W tile | A tile] [stats]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wv % RW, cw = wv / RW;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;

    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, p.w_bytes, 0x00020000);
    const int d_row = lane >> 3;
    const unsigned d_chunk_bytes = (unsigned)(((lane & 7) ^ d_row) << 4);
    // DMA instruction q of a stage (8 rows each): q < 48 weight rows 8 q .. of the current column half, then activation rows
    unsigned s_voff[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int q = wv * DPW + j;
        if (q < BNH / 8) {
            const int n = q * 8 + d_row;
            s_voff[j] = (unsigned)n * (unsigned)(p.ldw * ESZ) + d_chunk_bytes;
        } else {
            const int m = m0 + (q - BNH / 8) * 8 + d_row;
            s_voff[j] = (q < C::DMA_PER_STAGE && m < p.M) ? (unsigned)m * (unsigned)(p.lda * ESZ) + d_chunk_bytes : OOB_OFFSET;
        }
    }
    const unsigned half_bytes = (unsigned)BNH * (unsigned)(p.ldw * ESZ);  // weight rows of the second column half
    auto stage = [&](int half, int kt, int buf) {
        char* dst = smem + buf * STAGE + wv * DPW * 1024;
        const unsigned kb = (unsigned)(kt * BK * ESZ);
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int q = wv * DPW + j;
            if (q >= C::DMA_PER_STAGE) continue;  // (wave-uniform: the last wave of a 62-instruction stage has fewer)
            const unsigned vo = s_voff[j] == OOB_OFFSET ? OOB_OFFSET : s_voff[j] + kb;
            if (q < BNH / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + j * 1024), 16, vo + (unsigned)half * half_bytes, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };

    // K-steps are walked in a per-workgroup rotation: every CU streams the same weight tiles, in lockstep they would
    // all hit the same L2 lines at once.
    const int nk = p.K / BK;
    // (rank inside the XCD, blockIdx.x >> 3: the CUs behind one L2 then cover every rotation; blockIdx.x % nk gives an XCD's
    // 32 CUs only nk / gcd(nk, 8) distinct ones)
    const int k_rot = (int)(blockIdx.x >> 3) % nk;
    auto kstep = [&](int i) { const int k = i + k_rot; return k >= nk ? k - nk : k; };
    stage(0, kstep(0), 0);

    // The accumulators start from residual + bias: those HBM reads fly under the first DMA stage and the whole
    // K-loop instead of stalling the epilogue (lane layout: columns 384 half + 16 NFW cw + 16 nf + 4 f_kg + (0..3) of rows
    // 16 RFW rw + 16 rf + f_row).
    const int m_lane = m0 + rw * (16 * RFW) + f_row;  // + 16 rf: this lane's rows (kept as ONE register: row offsets and
    auto valid = [&](int rf) { return m_lane + rf * 16 < p.M; };          // validity are recomputed where they are used - the
    auto xrow = [&](int rf) { return (size_t)(m_lane + rf * 16) * BN; };  // 112 x 768 form has no registers to spare)
    f32x4 acc[NH][RFW][NFW];
#pragma unroll
    for (int rf = 0; rf < RFW; ++rf) {
        const int m = m_lane + rf * 16;
        const size_t rrow = p.res_mod > 0 ? (size_t)(m % p.res_mod) * BN : xrow(rf);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) {
                const int n = h * BNH + cw * (16 * NFW) + nf * 16 + f_kg * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.residual && valid(rf)) v += *reinterpret_cast<const f32x4*>(p.residual + rrow + n);
                acc[h][rf][nf] = v * p.w_s;
            }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = (h * nk + kt) & 1;
            if (kt + 1 < nk) stage(h, kstep(kt + 1), buf ^ 1);
            else if (h + 1 < NH) stage(h + 1, kstep(0), buf ^ 1);
            const char* wbase = smem + buf * STAGE;
            const char* abase = wbase + W_TILE;
            if constexpr (__is_same(T, SplitH) && (RFW > NFW)) {
                // tall wave tiles (112 x 48): hold the weight fragments, walk the row fragments (fewer live registers)
                f16x8 fwh[NFW], fwl[NFW];
#pragma unroll
                for (int nf = 0; nf < NFW; ++nf) {
                    fwh[nf] = *reinterpret_cast<const f16x8*>(wbase + swz(cw * (16 * NFW) + nf * 16 + f_row, f_kg));
                    fwl[nf] = *reinterpret_cast<const f16x8*>(wbase + swz(cw * (16 * NFW) + nf * 16 + f_row, 4 + f_kg));
                }
#pragma unroll
                for (int rf = 0; rf < RFW; ++rf) {
                    const f16x8 fah = *reinterpret_cast<const f16x8*>(abase + swz(rw * (16 * RFW) + rf * 16 + f_row, f_kg));
                    const f16x8 fal = *reinterpret_cast<const f16x8*>(abase + swz(rw * (16 * RFW) + rf * 16 + f_row, 4 + f_kg));
#pragma unroll
                    for (int nf = 0; nf < NFW; ++nf) acc[h][rf][nf] = split_mma(fwh[nf], fwl[nf], fah, fal, acc[h][rf][nf]);
                }
            } else if constexpr (__is_same(T, SplitH)) {
                f16x8 fah[RFW], fal[RFW];
#pragma unroll
                for (int rf = 0; rf < RFW; ++rf) {
                    fah[rf] = *reinterpret_cast<const f16x8*>(abase + swz(rw * (16 * RFW) + rf * 16 + f_row, f_kg));
                    fal[rf] = *reinterpret_cast<const f16x8*>(abase + swz(rw * (16 * RFW) + rf * 16 + f_row, 4 + f_kg));
                }
#pragma unroll
                for (int nf = 0; nf < NFW; ++nf) {
                    const f16x8 fwh = *reinterpret_cast<const f16x8*>(wbase + swz(cw * (16 * NFW) + nf * 16 + f_row, f_kg));
                    const f16x8 fwl = *reinterpret_cast<const f16x8*>(wbase + swz(cw * (16 * NFW) + nf * 16 + f_row, 4 + f_kg));
#pragma unroll
                    for (int rf = 0; rf < RFW; ++rf) acc[h][rf][nf] = split_mma(fwh, fwl, fah[rf], fal[rf], acc[h][rf][nf]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 fa[RFW];
#pragma unroll
                    for (int rf = 0; rf < RFW; ++rf) fa[rf] = *reinterpret_cast<const u32x4*>(abase + swz(rw * (16 * RFW) + rf * 16 + f_row, ks * 4 + f_kg));
#pragma unroll
                    for (int nf = 0; nf < NFW; ++nf) {
                        const u32x4 fw = *reinterpret_cast<const u32x4*>(wbase + swz(cw * (16 * NFW) + nf * 16 + f_row, ks * 4 + f_kg));
#pragma unroll
                        for (int rf = 0; rf < RFW; ++rf) acc[h][rf][nf] = mma(fw, fa[rf], acc[h][rf][nf], T{});
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: row statistics straight from the accumulators; a row's BN values sit in 4 lane groups x CW waves (x NH halves)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int rf = 0; rf < RFW; ++rf)
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) acc[h][rf][nf] *= p.w_inv;  // (the products carried the weights' power-of-two scale: exact)
    float* stat = reinterpret_cast<float*>(smem + 2 * STAGE);  // [2 passes][CW column groups][BM rows]
    float mean[RFW], rstd[RFW];
#pragma unroll
    for (int rf = 0; rf < RFW; ++rf) {
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) {
                const f32x4 v = acc[h][rf][nf];
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (f_kg == 0) stat[cw * BM + rw * (16 * RFW) + rf * 16 + f_row] = s;
    }
    __syncthreads();
#pragma unroll
    for (int rf = 0; rf < RFW; ++rf) {
        const int r = rw * (16 * RFW) + rf * 16 + f_row;
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) sm += stat[c * BM + r];
        mean[rf] = sm * (1.0f / BN);
        float q = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = acc[h][rf][nf][j] - mean[rf];
                    q = __builtin_fmaf(d, d, q);
                }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (f_kg == 0) stat[(CW + cw) * BM + r] = q;
    }
    __syncthreads();
#pragma unroll
    for (int rf = 0; rf < RFW; ++rf) {
        const int r = rw * (16 * RFW) + rf * 16 + f_row;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) var += stat[(CW + c) * BM + r];
        rstd[rf] = 1.0f / sqrtf(var * (1.0f / BN) + p.eps);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
            const int n = h * BNH + cw * (16 * NFW) + nf * 16 + f_kg * 4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + n), b = *reinterpret_cast<const f32x4*>(p.beta + n);
#pragma unroll
            for (int rf = 0; rf < RFW; ++rf) {
                if (!valid(rf)) continue;
                const f32x4 v = acc[h][rf][nf];
                const size_t off = xrow(rf) + n;
                *reinterpret_cast<f32x4*>(p.x_out + off) = v;
                f32x4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (v[j] - mean[rf]) * rstd[rf] * g[j] + b[j];
                if (p.h_bf16 == 2) {  // PP_OUT_SPLIT
                    split_store4(p.h_out, off, hv);
                } else if (p.h_bf16) {
                    const bf16x4 ho = {(__bf16)hv[0], (__bf16)hv[1], (__bf16)hv[2], (__bf16)hv[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.h_out) + off) = ho;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.h_out) + off) = hv;
                }
            }
        }
}

template <typename T, typename C>
static int launch(const Params& p, hipStream_t s) {
    auto kern = gemm_res_ln_kernel<T, C>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    hipLaunchKernelGGL(kern, dim3((p.M + C::BM - 1) / C::BM), dim3(C::THREADS), C::LDS, s, p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

}  // namespace rl
}  // namespace pp

extern "C" int pp_gemm_residual_layernorm(int prec, const void* act, const void* weight, const float* bias,
                                          const float* residual, int res_mod, float* x_out, const float* gamma,
                                          const float* beta, float eps, void* h_out, int h_bf16, int M, int N, int K,
                                          int lda, int ldw, void* stream) {
    return pp_gemm_residual_layernorm_ws(prec, act, weight, bias, residual, res_mod, x_out, gamma, beta, eps, h_out, h_bf16, M, N, K, lda, ldw, 1.0f,
                                         stream);
}

extern "C" int pp_gemm_residual_layernorm_ws(int prec, const void* act, const void* weight, const float* bias,
                                             const float* residual, int res_mod, float* x_out, const float* gamma,
                                             const float* beta, float eps, void* h_out, int h_bf16, int M, int N, int K,
                                             int lda, int ldw, float w_inv_scale, void* stream) {
    using namespace pp;
    {
        unsigned u;
        __builtin_memcpy(&u, &w_inv_scale, 4);
        PP_REQUIRE((u >> 31) == 0 && (u & 0x007fffffu) == 0 && ((u >> 23) & 0xffu) >= 127 - 40 && ((u >> 23) & 0xffu) <= 127 + 40, PP_ERR_INVALID_ARG,
                   "pp_gemm_residual_layernorm: the weight scale must be a power of two in [2^-40, 2^40]");
        PP_REQUIRE(w_inv_scale == 1.0f || prec == PP_PREC_F16X3, PP_ERR_INVALID_ARG, "pp_gemm_residual_layernorm: weight scales belong to the split-fp16 mode");
    }
    PP_REQUIRE(act && weight && x_out && gamma && beta && h_out, PP_ERR_INVALID_ARG,
               "pp_gemm_residual_layernorm: NULL argument");
    PP_REQUIRE(N == rl::C384::BN || N == rl::C768::BN, PP_ERR_UNSUPPORTED,
               "pp_gemm_residual_layernorm: the fused kernel is built for N = 384 and N = 768");
    PP_REQUIRE(M > 0 && K > 0, PP_ERR_INVALID_ARG, "pp_gemm_residual_layernorm: M and K must be positive");
    const int bk = prec == PP_PREC_BF16 ? 64 : 32;
    const size_t esz = prec == PP_PREC_BF16 ? 2 : 4;
    PP_REQUIRE(prec == PP_PREC_BF16 || prec == PP_PREC_F32 || prec == PP_PREC_F16X3, PP_ERR_INVALID_ARG,
               "pp_gemm_residual_layernorm: unknown precision");
    PP_REQUIRE(h_bf16 == PP_OUT_F32 || h_bf16 == (prec == PP_PREC_BF16 ? PP_OUT_BF16 : (prec == PP_PREC_F16X3 ? PP_OUT_SPLIT : PP_OUT_F32)),
               PP_ERR_UNSUPPORTED, "pp_gemm_residual_layernorm: h_out is fp32 or the operand format of the precision mode");
    PP_REQUIRE(prec != PP_PREC_F16X3 || (lda % 32 == 0 && ldw % 32 == 0), PP_ERR_UNSUPPORTED,
               "pp_gemm_residual_layernorm: split-fp16 operands need row pitches that are multiples of 32 elements");
    PP_REQUIRE(K % bk == 0 && lda % 8 == 0 && ldw % 8 == 0, PP_ERR_UNSUPPORTED,
               "pp_gemm_residual_layernorm: K must be a multiple of the K-tile, lda/ldw multiples of 8");
    const size_t ab = ((size_t)(M - 1) * lda + K) * esz, wb = ((size_t)(N - 1) * ldw + K) * esz;
    PP_REQUIRE(ab < rl::OOB_OFFSET && wb < rl::OOB_OFFSET, PP_ERR_UNSUPPORTED,
               "pp_gemm_residual_layernorm: operands must be smaller than 2 GiB");
    rl::Params p{};
    p.A = act; p.W = weight; p.bias = bias; p.residual = residual; p.x_out = x_out; p.h_out = h_out;
    p.gamma = gamma; p.beta = beta; p.M = M; p.K = K; p.lda = lda; p.ldw = ldw; p.res_mod = res_mod;
    p.h_bf16 = h_bf16; p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.eps = eps;
    p.w_inv = w_inv_scale; p.w_s = 1.0f / w_inv_scale;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool wide = N == rl::C768::BN;
    if (prec == PP_PREC_BF16) return wide ? rl::launch<__bf16, rl::C768>(p, s) : rl::launch<__bf16, rl::C384>(p, s);
    if (prec == PP_PREC_F16X3) return wide ? rl::launch<SplitH, rl::C768>(p, s) : rl::launch<SplitH, rl::C384>(p, s);
    return wide ? rl::launch<float, rl::C768>(p, s) : rl::launch<float, rl::C384>(p, s);
}

