// Fused ViT feed-forward block for gfx950 (bf16 operands):
//     x <- x + GELU(h W1^T + b1) W2^T + b2 ;   h_next <- LayerNorm(x)
// (mmpretrain TransformerEncoderLayer [3P]: x = ffn(ln2(x), identity = x), FFN = Linear - GELU(erf) - Linear, followed by
// the next layer's ln1 or the final ln1). Done as two GEMM launches, the 4x-wide hidden activation (75 MB at bs 64)
// is written to HBM and read back; it is a quarter of all the bytes a ViT-S layer moves. Here it never leaves the CU:
//
//   * one workgroup owns 96 complete token rows (grid = M / 96 = one workgroup per CU at bs 64 with flip test),
//     768 threads = 12 waves, wave (rw, cw): rows 16 rw .. +15, column half cw;
//   * the LayerNorm-ed input rows h live in REGISTERS as MFMA operand fragments for the whole kernel (48 VGPRs);
//   * the hidden layer is processed in 12 chunks of 128 units:
//       phase A  P = h W1[chunk]^T       6 K-steps, W1 tile 128 x 64 streamed by LDS-DMA (16 KiB stages)
//                G = GELU(P + b1) -> bf16 -> LDS as the operand tile of phase B (24 KiB)
//       phase B  acc += G W2[:, chunk]^T  2 K-steps, W2 tile 384 x 64 streamed by LDS-DMA (48 KiB stages)
//     with the next stage's DMA always in flight under the current stage's MFMAs;
//   * the 96 x 384 output accumulators start from residual + b2 and end in the same LayerNorm epilogue as
//     pp_gemm_ln.hip (row statistics in registers, one LDS exchange between the column halves).
// GELU uses erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, well under bf16 resolution): libdevice erff costs as
// many VALU cycles as the MFMAs of the whole block.
#include "pp_common.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace mlp {

constexpr int BM = 96, E = 384, CHUNK = 128;
constexpr int WAVES = 12, THREADS = 64 * WAVES;
constexpr int ROW_BYTES = 128, BK = 64;
constexpr int W1_STAGE = CHUNK * ROW_BYTES;       // 16 KiB: 128 hidden units x 64 k
constexpr int W2_STAGE = E * ROW_BYTES;           // 48 KiB: 384 outputs x 64 hidden units
constexpr int HS_TILE = BM * ROW_BYTES;           // 12 KiB per 64 hidden units
constexpr int OFF_W1 = 0;
constexpr int OFF_W2 = 2 * W1_STAGE;
constexpr int OFF_HS = OFF_W2 + 2 * W2_STAGE;
constexpr int OFF_STAT = OFF_HS + 2 * HS_TILE;
constexpr int LDS = OFF_STAT + 4 * BM * 4;        // 155 136 B
constexpr int KT1 = E / BK;                       // 6 K-steps in phase A

struct Params {
    const __bf16* h;       // [M, 384] LayerNorm-ed block input
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
    unsigned w1_bytes, w2_bytes;
    float eps;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)), erf by A&S 7.1.26
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = __builtin_fmaf(-poly, e, 1.0f);
    const float erf = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf);
}

__global__ __launch_bounds__(THREADS, 3) void mlp_res_ln_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wv % 6, cw = wv / 6;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int m = m0 + rw * 16 + f_row;
    const bool valid = m < p.M;
    const size_t xrow = (size_t)m * E;

    const __amdgpu_buffer_rsrc_t w1_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.W1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.W2), 0, p.w2_bytes, 0x00020000);
    const int d_row = lane >> 3;
    const unsigned d_chunk_bytes = (unsigned)(((lane & 7) ^ d_row) << 4);

    // W1 tile (chunk c, K-step kt): rows 128 c + r, bytes [128 kt, +128) -> 16 DMA instructions, waves 0..7 two each
    auto stage_w1 = [&](int c, int kt, int slot) {
        if (wv < 8) {
            char* dst = smem + OFF_W1 + slot * W1_STAGE + wv * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (wv * 2 + j) * 8 + d_row;
                const unsigned vo = (unsigned)(c * CHUNK + r) * (unsigned)(E * 2) + (unsigned)(kt * 128) + d_chunk_bytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rsrc, (lds_ptr_t)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        }
    };
    // W2 tile (chunk c, K-step k2): rows n = 0..383, bytes [(128 c + 64 k2) * 2, +128); `half` selects rows [192 half, +192):
    // 24 DMA instructions per half, two per wave
    auto stage_w2_half = [&](int c, int k2, int slot, int half) {
        char* dst = smem + OFF_W2 + slot * W2_STAGE + half * (W2_STAGE / 2) + wv * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = half * 192 + (wv * 2 + j) * 8 + d_row;
            const unsigned vo = (unsigned)n * (unsigned)(p.F * 2) + (unsigned)((c * CHUNK + k2 * BK) * 2) + d_chunk_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (lds_ptr_t)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };

    // Every workgroup walks the hidden chunks in a different rotation: all 256 CUs stream the SAME weights, and in
    // lockstep they would all hit the same L2 lines at the same moment.
    const int nchunks = p.F / CHUNK;
    const int c_rot = blockIdx.x % nchunks;
    auto chunk_of = [&](int i) { const int c = i + c_rot; return c >= nchunks ? c - nchunks : c; };

    // ---- prologue: first stages in flight, input rows into registers, accumulators = residual + b2
    stage_w1(chunk_of(0), 0, 0);
    stage_w2_half(chunk_of(0), 0, 0, 0);
    stage_w2_half(chunk_of(0), 0, 0, 1);
    u32x4 hf[KT1][2];
    {
        const __bf16* hrow = p.h + (size_t)(valid ? m : 0) * E;
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                hf[kt][ks] = *reinterpret_cast<const u32x4*>(hrow + kt * BK + (ks * 4 + f_kg) * 8);
    }
    f32x4 acc[12];
#pragma unroll
    for (int nf = 0; nf < 12; ++nf) {
        const int n = cw * 192 + nf * 16 + f_kg * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(p.b2 + n);
        if (valid) v += *reinterpret_cast<const f32x4*>(p.residual + xrow + n);
        acc[nf] = v;
    }
    __syncthreads();

    int a_it = 0;  // phase-A stage counter: W1 ring slot = a_it & 1
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = chunk_of(ci);
        const int c_next = chunk_of(ci + 1 < nchunks ? ci + 1 : ci);
        // ================= phase A: P[16 rows x 64 hidden] = h W1[chunk]^T
        f32x4 pacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt, ++a_it) {
            const int slot = a_it & 1;
            if (kt + 1 < KT1) stage_w1(c, kt + 1, slot ^ 1);
            if (kt == KT1 - 2) stage_w2_half(c, 1, 1, 0);  // second W2 K-step of this chunk, spread over two A steps
            if (kt == KT1 - 1) stage_w2_half(c, 1, 1, 1);
            const char* wbase = smem + OFF_W1 + slot * W1_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    const u32x4 fw = *reinterpret_cast<const u32x4*>(wbase + swz(cw * 64 + nf * 16 + f_row, ks * 4 + f_kg));
                    pacc[nf] = mma(fw, hf[kt][ks], pacc[nf]);
                }
            if (kt + 1 < KT1) __syncthreads();
        }
        // GELU -> bf16 -> operand tile of phase B. Lane holds hidden units 64 cw + 16 nf + 4 f_kg + (0..3) of its row:
        // K-step cw of the chunk, 16-byte chunk 2 nf + (f_kg >> 1), upper or lower 8 bytes.
        {
            char* hs = smem + OFF_HS + cw * HS_TILE;
            const int r = rw * 16 + f_row;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b1 + c * CHUNK + cw * 64 + nf * 16 + f_kg * 4);
                const f32x4 v = pacc[nf] + bv;
                const bf16x4 g = {(__bf16)gelu_fast(v[0]), (__bf16)gelu_fast(v[1]), (__bf16)gelu_fast(v[2]),
                                  (__bf16)gelu_fast(v[3])};
                *reinterpret_cast<bf16x4*>(hs + swz(r, 2 * nf + (f_kg >> 1)) + (f_kg & 1) * 8) = g;
            }
        }
        __syncthreads();  // G complete; W2 K-step 1 landed; W1 ring free

        // ================= phase B: acc[16 rows x 192 cols] += G W2[:, chunk]^T
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if (ci + 1 < nchunks) {
                if (k2 == 0) stage_w1(c_next, 0, a_it & 1);                    // next chunk's first W1 tile
                else { stage_w2_half(c_next, 0, 0, 0); stage_w2_half(c_next, 0, 0, 1); }  // and its first W2 tile
            }
            const char* wbase = smem + OFF_W2 + k2 * W2_STAGE;
            const char* hbase = smem + OFF_HS + k2 * HS_TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 fa = *reinterpret_cast<const u32x4*>(hbase + swz(rw * 16 + f_row, ks * 4 + f_kg));
#pragma unroll
                for (int nf = 0; nf < 12; ++nf) {
                    const u32x4 fw = *reinterpret_cast<const u32x4*>(wbase + swz(cw * 192 + nf * 16 + f_row, ks * 4 + f_kg));
                    acc[nf] = mma(fw, fa, acc[nf]);
                }
            }
            __syncthreads();
        }
    }

    // ---- LayerNorm epilogue (same as pp_gemm_ln.hip)
    float s = 0.f;
#pragma unroll
    for (int nf = 0; nf < 12; ++nf) {
        const f32x4 v = acc[nf];
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    float* stat = reinterpret_cast<float*>(smem + OFF_STAT);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (f_kg == 0) stat[cw * BM + rw * 16 + f_row] = s;
    __syncthreads();
    const float mean = (stat[rw * 16 + f_row] + stat[BM + rw * 16 + f_row]) * (1.0f / E);
    float q = 0.f;
#pragma unroll
    for (int nf = 0; nf < 12; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = acc[nf][j] - mean;
            q = __builtin_fmaf(d, d, q);
        }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (f_kg == 0) stat[2 * BM + cw * BM + rw * 16 + f_row] = q;
    __syncthreads();
    const float var = (stat[2 * BM + rw * 16 + f_row] + stat[3 * BM + rw * 16 + f_row]) * (1.0f / E);
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (!valid) return;
#pragma unroll
    for (int nf = 0; nf < 12; ++nf) {
        const int n = cw * 192 + nf * 16 + f_kg * 4;
        const f32x4 v = acc[nf];
        *reinterpret_cast<f32x4*>(p.x_out + xrow + n) = v;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + n), b = *reinterpret_cast<const f32x4*>(p.beta + n);
        const bf16x4 hv = {(__bf16)((v[0] - mean) * rstd * g[0] + b[0]), (__bf16)((v[1] - mean) * rstd * g[1] + b[1]),
                           (__bf16)((v[2] - mean) * rstd * g[2] + b[2]), (__bf16)((v[3] - mean) * rstd * g[3] + b[3])};
        *reinterpret_cast<bf16x4*>(p.h_out + xrow + n) = hv;
    }
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
    PP_REQUIRE((size_t)F * E * 2 < 0x7ffffff0u, PP_ERR_UNSUPPORTED, "pp_mlp_residual_layernorm: weights exceed 2 GiB");
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
    p.w1_bytes = (unsigned)((size_t)F * E * 2);
    p.w2_bytes = (unsigned)((size_t)E * F * 2);
    p.eps = eps;
    auto kern = mlp::mlp_res_ln_kernel;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, mlp::LDS));
    hipLaunchKernelGGL(kern, dim3((M + mlp::BM - 1) / mlp::BM), dim3(mlp::THREADS), mlp::LDS,
                       reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
