// Host side of the fused ViT feed-forward launch for gfx950 in the parity precision (PP_PREC_F16X3: split-fp16 operands, pp_split.h, three
// fp16 MFMAs per product):
//     [x <- x + att Wp^T + bp ; h <- LN2(x)]   x <- x + GELU(h W1^T + b1) W2^T + b2 ;   h_next <- LayerNorm(x)
// (mmpretrain TransformerEncoderLayer [3P]: x = ffn(ln2(x), identity = x), FFN = Linear - GELU(erf) - Linear, then the next
// layer's ln1 / the final ln1; call site mmpose/models/pose_estimators/base.py:206, ctor args
// configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67).
// This file holds the geometry of the packed weight streams, the kernels that pack them, and the C entry points; the kernel itself is
// pp_ffn_dma.hip (eight computing waves + four DMA-only waves on 96 complete token rows per workgroup). The eight-wave, role-alternating kernel
// of round 3 that used to live here was retired in round 6 (the twelve-wave form replaced it in round 4; git history has it).
//
//   * the hidden layer runs in chunks of 128 units; per chunk twenty steps on a ring of FOUR 28 KiB slots:
//       A-step kb (12 per chunk)   P += x[:, kb] W1[chunk, kb]^T   slot = W1 block (128 lines x 128 B) + x block (96 lines)
//       B-step (j, half) (8)       acc[:, half] += G[:, j] W2[half, chunk j]^T   slot = W2 half block (192 lines)
//   * W1 / W2 come PRE-PACKED in consumption order (pp_ffn_split_pack_weights: per chunk 12 W1 blocks of 16 KiB, then
//     8 W2 half blocks of 24 KiB, 128-byte lines with the LDS XOR swizzle already applied), so a weight DMA instruction
//     is a linear 1 KiB copy; the x lines are 128-byte segments of the row-major split tensor, swizzled at the source.
// LDS: 48 KiB G + 4 x 28 KiB ring = 160 KiB.
#include "pp_common.h"
#include "pp_split.h"
#include "pp_ffn_params.h"

namespace pp {
namespace ffs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

[[maybe_unused]] constexpr int BM = 96, E = 384, CHUNK = 128, THREADS = 512;
constexpr int KB = E / 32;                       // 12 k-blocks of the input width
constexpr int G_KB = BM * 128;                   // 12 KiB: 96 rows x one 128-byte block
[[maybe_unused]] constexpr int OFF_G = 0;                         // [4][96][128 B]
constexpr int OFF_RING = 4 * G_KB;               // 48 KiB
constexpr int SLOTB = 28 * 1024, NSLOT = 4;
constexpr int LDS = OFF_RING + NSLOT * SLOTB;    // 163 840 B
[[maybe_unused]] constexpr int X_OFF = 16 * 1024;                 // A slot: the x lines sit behind the 128 weight lines
constexpr int NA = KB, NB = 8, STEPS = NA + NB;  // steps per chunk
constexpr int A_BLOCK = CHUNK * 128;             // 16 KiB
constexpr int B_BLOCK = (E / 2) * 128;           // 24 KiB
constexpr int B_PART = NA * A_BLOCK;             // 192 KiB: offset of the W2 half blocks inside a chunk's stream
constexpr int CHUNK_BYTES = B_PART + NB * B_BLOCK;  // 384 KiB
constexpr unsigned OOB = 0x7ffffff0u;
static_assert(LDS == 160 * 1024, "LDS map");
static_assert(NA % NSLOT == 0 && STEPS % NSLOT == 0, "ring positions must repeat per chunk");


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

int launch_dma_form(const Params& p, bool proj, hipStream_t s);  // pp_ffn_dma.hip
}  // namespace ffs
}  // namespace pp

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

// A weight scale is a positive power of two (weights.py stores w * 2^e and hands over 2^-e): anything else is refused - the kernels rely on the
// multiplications by it being exact.
static bool is_pow2_scale(float v) {
    unsigned u;
    static_assert(sizeof(u) == sizeof(v), "float bits");
    __builtin_memcpy(&u, &v, 4);
    const unsigned ex = (u >> 23) & 0xffu;
    return (u >> 31) == 0 && (u & 0x007fffffu) == 0 && ex >= 127 - 40 && ex <= 127 + 40;
}

static void fill_ffn(pp::ffs::Params& p, const void* h, const void* w_packed, const float* b1, const float* b2, const void* residual, float* x_out,
                     const float* gamma, const float* beta, float eps, void* h_out, int M, int E, int F, float w1_inv, float w2_inv) {
    using namespace pp;
    p.h = h;
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
    p.inv_1 = w1_inv; p.s_1 = 1.0f / w1_inv;
    p.inv_2 = w2_inv; p.s_2 = 1.0f / w2_inv;
    p.inv_p = p.s_p = 1.0f;
}

static void fill_proj(pp::ffs::Params& p, const void* att, const void* wproj_packed, const float* bproj, const float* gamma2, const float* beta2,
                      float wp_inv) {
    using namespace pp;
    p.att = att;
    p.wproj = wproj_packed;
    p.bp = bproj;
    p.gamma2 = gamma2;
    p.beta2 = beta2;
    p.att_bytes = p.h_bytes;
    p.wproj_bytes = (unsigned)(2 * ffs::KB * ffs::B_BLOCK);
    p.inv_p = wp_inv; p.s_p = 1.0f / wp_inv;
}

#define PP_FFN_SHAPE_CHECKS(name)                                                                                                          \
    PP_REQUIRE(E == ffs::E, PP_ERR_UNSUPPORTED, name ": built for embed dim 384 (ViT-S)");                                                 \
    PP_REQUIRE(M > 0 && F > 0 && F % ffs::CHUNK == 0, PP_ERR_UNSUPPORTED, name ": hidden width must be a positive multiple of 128");       \
    PP_REQUIRE((size_t)M * E * 4 < ffs::OOB && (size_t)(F / ffs::CHUNK) * ffs::CHUNK_BYTES < ffs::OOB, PP_ERR_UNSUPPORTED, name ": operand exceeds 2 GiB")

extern "C" int pp_ffn_split_residual_layernorm_ws(const void* h_in, const void* w_packed, const float* b1, const float* b2, const float* residual,
                                                  float* x_out, const float* gamma, const float* beta, float eps, void* h_out, int M, int E, int F,
                                                  float w1_inv_scale, float w2_inv_scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(h_in && w_packed && b1 && b2 && residual && x_out && gamma && beta && h_out, PP_ERR_INVALID_ARG,
               "pp_ffn_split_residual_layernorm: NULL argument");
    PP_FFN_SHAPE_CHECKS("pp_ffn_split_residual_layernorm");
    PP_REQUIRE(is_pow2_scale(w1_inv_scale) && is_pow2_scale(w2_inv_scale), PP_ERR_INVALID_ARG,
               "pp_ffn_split_residual_layernorm: a weight scale must be a power of two in [2^-40, 2^40]");
    ffs::Params p{};
    fill_ffn(p, h_in, w_packed, b1, b2, residual, x_out, gamma, beta, eps, h_out, M, E, F, w1_inv_scale, w2_inv_scale);
    return ffs::launch_dma_form(p, false, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_ffn_split_residual_layernorm(const void* h_in, const void* w_packed, const float* b1, const float* b2,
                                               const float* residual, float* x_out, const float* gamma, const float* beta,
                                               float eps, void* h_out, int M, int E, int F, void* stream) {
    return pp_ffn_split_residual_layernorm_ws(h_in, w_packed, b1, b2, residual, x_out, gamma, beta, eps, h_out, M, E, F, 1.0f, 1.0f, stream);
}

extern "C" int pp_proj_ffn_split_residual_layernorm_ws(const void* att, const void* wproj_packed, const float* bproj, const float* gamma2,
                                                       const float* beta2, void* h_scratch, const void* w_packed, const float* b1,
                                                       const float* b2, const float* residual, float* x_out, const float* gamma,
                                                       const float* beta, float eps, void* h_out, int M, int E, int F, float wp_inv_scale,
                                                       float w1_inv_scale, float w2_inv_scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(att && wproj_packed && bproj && gamma2 && beta2 && h_scratch && w_packed && b1 && b2 && residual && x_out && gamma &&
                   beta && h_out,
               PP_ERR_INVALID_ARG, "pp_proj_ffn_split_residual_layernorm: NULL argument");
    PP_FFN_SHAPE_CHECKS("pp_proj_ffn_split_residual_layernorm");
    PP_REQUIRE(h_scratch != att && h_scratch != h_out, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_residual_layernorm: h_scratch must not alias the attention rows or h_out");
    PP_REQUIRE(is_pow2_scale(wp_inv_scale) && is_pow2_scale(w1_inv_scale) && is_pow2_scale(w2_inv_scale), PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_residual_layernorm: a weight scale must be a power of two in [2^-40, 2^40]");
    ffs::Params p{};
    fill_ffn(p, h_scratch, w_packed, b1, b2, residual, x_out, gamma, beta, eps, h_out, M, E, F, w1_inv_scale, w2_inv_scale);
    fill_proj(p, att, wproj_packed, bproj, gamma2, beta2, wp_inv_scale);
    return ffs::launch_dma_form(p, true, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pp_proj_ffn_split_residual_layernorm(const void* att, const void* wproj_packed, const float* bproj,
                                                    const float* gamma2, const float* beta2, void* h_scratch,
                                                    const void* w_packed, const float* b1, const float* b2,
                                                    const float* residual, float* x_out, const float* gamma, const float* beta,
                                                    float eps, void* h_out, int M, int E, int F, void* stream) {
    return pp_proj_ffn_split_residual_layernorm_ws(att, wproj_packed, bproj, gamma2, beta2, h_scratch, w_packed, b1, b2, residual, x_out, gamma, beta,
                                                   eps, h_out, M, E, F, 1.0f, 1.0f, 1.0f, stream);
}

// The projection + FFN launch inside a chain of layers whose LayerNorm in front of the qkv projection is folded into that projection
// (pp_qkv_attention_split_folded). Rows that travel between the launches of the chain are CENTERED: h_out holds x - mean(x) per row in the operand
// format and stats_out (mean, rstd) - the next layer's projection then needs no mean * colsum correction (that difference of two fp32 numbers of the
// size of |mean| * |colsum| cost 2 - 5x the error of a plain LayerNorm at |mean| / std <= 1 and a digit more per decade of |mean| / std beyond), and
// this launch gets its residual back as  (hi + lo) + mean  from `residual` (PP_OUT_SPLIT) and `residual_stats`.
extern "C" int pp_proj_ffn_split_folded(const void* att, const void* wproj_packed, const float* bproj, const float* gamma2, const float* beta2,
                                        void* h_scratch, const void* w_packed, const float* b1, const float* b2, const void* residual,
                                        int residual_format, const float* residual_stats, int fold_out, float* x_out, const float* gamma,
                                        const float* beta, float eps, void* h_out, float* stats_out, int M, int E, int F, float wp_inv_scale,
                                        float w1_inv_scale, float w2_inv_scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(att && wproj_packed && bproj && gamma2 && beta2 && h_scratch && w_packed && b1 && b2 && residual && h_out, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: NULL argument");
    PP_REQUIRE(residual_format == PP_OUT_F32 || residual_format == PP_OUT_SPLIT, PP_ERR_INVALID_ARG, "pp_proj_ffn_split_folded: residual_format is PP_OUT_F32 or PP_OUT_SPLIT");
    PP_REQUIRE(residual_format != PP_OUT_SPLIT || residual_stats, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: operand-format residual rows are centered rows: residual_stats (their means) is required");
    PP_REQUIRE(fold_out ? stats_out != nullptr : (x_out && gamma && beta), PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: fold_out needs stats_out; without it x_out, gamma and beta are required");
    PP_FFN_SHAPE_CHECKS("pp_proj_ffn_split_folded");
    PP_REQUIRE((F / ffs::CHUNK) % 2 == 0, PP_ERR_UNSUPPORTED,
               "pp_proj_ffn_split_folded: hidden width must be an even number of 128-column chunks (the paired twelve-wave kernel)");
    PP_REQUIRE(h_scratch != att && h_scratch != h_out && h_scratch != residual, PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: h_scratch must not alias the attention rows, the residual rows or h_out");
    PP_REQUIRE(is_pow2_scale(wp_inv_scale) && is_pow2_scale(w1_inv_scale) && is_pow2_scale(w2_inv_scale), PP_ERR_INVALID_ARG,
               "pp_proj_ffn_split_folded: a weight scale must be a power of two in [2^-40, 2^40]");
    ffs::Params p{};
    fill_ffn(p, h_scratch, w_packed, b1, b2, residual, x_out, gamma, beta, eps, h_out, M, E, F, w1_inv_scale, w2_inv_scale);
    fill_proj(p, att, wproj_packed, bproj, gamma2, beta2, wp_inv_scale);
    p.res_split = residual_format == PP_OUT_SPLIT;
    p.res_stats = residual_stats;
    p.fold_out = fold_out != 0;
    p.stats_out = stats_out;
    return ffs::launch_dma_form(p, true, reinterpret_cast<hipStream_t>(stream));
}
