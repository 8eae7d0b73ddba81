// Memory-bound pieces of the ViT backbone for gfx950: input preprocessing fused with the
// patch-embed im2col, and LayerNorm as a wavefront reduction.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// uint8 BGR CHW crops -> patch matrix for the patch-embed GEMM, both flip-test passes at once.
//   row  m = (pass * B + b) * (Hp * Wp) + py * Wp + px          (pass 1 = horizontally flipped crop)
//   col  k = c * P * P + i * P + j                               (= Conv2d weight.flatten(1) order)
//   A[m, k] = (img[b, src_c, P py + i - pad, xs] - mean[c]) / std[c],  0 outside the image (the conv's
//   zero padding acts on the NORMALISED image), src_c = 2 - c when bgr_to_rgb,
//   xs = P px + j - pad for pass 0 and W - 1 - (P px + j - pad) for pass 1.
// One thread per (m, c, i): 16 source bytes -> 16 outputs.
// TIn = uint8_t: raw crops, normalised here; TIn = float: an already preprocessed (B,3,H,W) tensor
// (the reference backbone's own input contract), copied as is.
template <typename T, typename TIn>
__global__ __launch_bounds__(256) void preproc_im2col_kernel(const TIn* __restrict__ img, T* __restrict__ A, int B,
                                                             int passes, int H, int W, int Hp, int Wp, int pad,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             int bgr_to_rgb) {
    constexpr int P = 16;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)passes * B * Hp * Wp * 3 * P;
    if (gid >= total) return;
    const int i = (int)(gid % P);
    const int c = (int)((gid / P) % 3);
    const long long m = gid / (3 * P);
    const int np = Hp * Wp;
    const int pb = (int)(m / np), pp_ = (int)(m - (long long)pb * np);
    const int pass = pb / B, b = pb - pass * B;
    const int py = pp_ / Wp, px = pp_ - py * Wp;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const int sc = (sizeof(TIn) == 1 && bgr_to_rgb) ? 2 - c : c;
    const int y = P * py + i - pad;
    float v[P];
    const int x0 = P * px - pad;  // first source column of this patch row (pass 0 order)
    bool fast = false;
    if constexpr (sizeof(TIn) == 1) {
        // uint8 crops, interior patch rows: the 16 source bytes [x0, x0 + 16) (pass 1: the mirrored range, read
        // backwards) sit inside five aligned dwords - five loads instead of sixteen byte loads
        fast = y >= 0 && y < H && x0 >= 2 && x0 + P + 2 <= W && ((x0 - 2) & 3) == 0 && (W & 3) == 0;
        if (fast) {
            const int start = pass ? W - 1 - (x0 + P - 1) : x0;  // lowest source column; start - 2 is 4-byte aligned
            const uint32_t* src = reinterpret_cast<const uint32_t*>(img + (((size_t)b * 3 + sc) * H + y) * W + start - 2);
            uint32_t w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[q] = src[q];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int pos = pass ? 2 + (P - 1 - j) : 2 + j;  // byte position inside the 20 loaded bytes
                const float u = (float)((w[pos >> 2] >> ((pos & 3) * 8)) & 0xffu);
                v[j] = (u - mean) / stdv;
            }
        }
    }
    if (!fast) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int x = x0 + j;
            const int xs = pass ? W - 1 - x : x;
            float val = 0.f;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const float u = (float)img[(((size_t)b * 3 + sc) * H + y) * W + xs];
                val = sizeof(TIn) == 1 ? (u - mean) / stdv : u;
            }
            v[j] = val;
        }
    }
    T* dst = A + (size_t)m * (3 * P * P) + c * P * P + i * P;
    if constexpr (__is_same(T, SplitH)) {  // 16 consecutive k = half a 32-element block: 32 B of hi halves, 32 B of lo halves
        f16x8 h0, h1, l0, l1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h0[j] = split_hi(v[j]);
            l0[j] = split_lo(v[j], h0[j]);
            h1[j] = split_hi(v[8 + j]);
            l1[j] = split_lo(v[8 + j], h1[j]);
        }
        char* o = split_addr(A, (size_t)m * (3 * P * P) + c * P * P + i * P);
        reinterpret_cast<f16x8*>(o)[0] = h0;
        reinterpret_cast<f16x8*>(o)[1] = h1;
        reinterpret_cast<f16x8*>(o + 64)[0] = l0;
        reinterpret_cast<f16x8*>(o + 64)[1] = l1;
    } else if constexpr (sizeof(T) == 2) {
        bf16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o0[j] = (__bf16)v[j];
            o1[j] = (__bf16)v[8 + j];
        }
        reinterpret_cast<bf16x8*>(dst)[0] = o0;
        reinterpret_cast<bf16x8*>(dst)[1] = o1;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(dst)[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim of an fp32 [M, E] matrix, one wavefront per row, fp32 statistics
// (two-pass: mean, then centred variance), output bf16 or fp32. E % 128 == 0, E <= 1024.
template <typename TO, int EV>  // EV = E / 128 float2 pairs... per lane: EV * 2 floats
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ y, int M,
                                                        float eps) {
    constexpr int E = EV * 128;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x2* xr = reinterpret_cast<const f32x2*>(x + (size_t)row * E);
    f32x2 v[EV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EV; ++i) {
        v[i] = xr[lane + 64 * i];
        s += v[i][0] + v[i][1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.0f / E);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EV; ++i) {
        const float a = v[i][0] - mean, b = v[i][1] - mean;
        q = __builtin_fmaf(a, a, q);
        q = __builtin_fmaf(b, b, q);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q * (1.0f / E) + eps);
    const f32x2* g2 = reinterpret_cast<const f32x2*>(gamma);
    const f32x2* b2 = reinterpret_cast<const f32x2*>(beta);
#pragma unroll
    for (int i = 0; i < EV; ++i) {
        const f32x2 g = g2[lane + 64 * i], b = b2[lane + 64 * i];
        const float o0 = (v[i][0] - mean) * rstd * g[0] + b[0];
        const float o1 = (v[i][1] - mean) * rstd * g[1] + b[1];
        if constexpr (__is_same(TO, SplitH)) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const _Float16 h0 = split_hi(o0), h1 = split_hi(o1);
            char* o = split_addr(y, (size_t)row * E + 2 * (lane + 64 * i));
            *reinterpret_cast<f16x2*>(o) = f16x2{h0, h1};
            *reinterpret_cast<f16x2*>(o + 64) = f16x2{split_lo(o0, h0), split_lo(o1, h1)};
        } else if constexpr (sizeof(TO) == 2) {
            reinterpret_cast<bf16x2*>(y + (size_t)row * E)[lane + 64 * i] = bf16x2{(__bf16)o0, (__bf16)o1};
        } else {
            reinterpret_cast<f32x2*>(y + (size_t)row * E)[lane + 64 * i] = f32x2{o0, o1};
        }
    }
}

template <typename TO>
static int launch_ln(const float* x, const float* g, const float* b, void* y, int M, int E, float eps, hipStream_t s) {
    const dim3 grid((M + 3) / 4), block(256);
    TO* yo = reinterpret_cast<TO*>(y);
    switch (E) {
        case 384: hipLaunchKernelGGL((layernorm_kernel<TO, 3>), grid, block, 0, s, x, g, b, yo, M, eps); break;
        case 768: hipLaunchKernelGGL((layernorm_kernel<TO, 6>), grid, block, 0, s, x, g, b, yo, M, eps); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<TO, 8>), grid, block, 0, s, x, g, b, yo, M, eps); break;
        default: return fail(PP_ERR_UNSUPPORTED, "pp_layernorm: embed dim must be 384, 768 or 1024");
    }
    PP_LAUNCH_CHECK_AS("layernorm");
    return PP_OK;
}

}  // namespace pp

extern "C" int pp_preproc_im2col(int prec, const void* img, int img_is_f32, void* patches, int B, int passes, int H,
                                 int W, int patch, int pad, const float* mean_host, const float* std_host,
                                 int bgr_to_rgb, void* stream) {
    using namespace pp;
    PP_REQUIRE(img && patches, PP_ERR_INVALID_ARG, "pp_preproc_im2col: NULL argument");
    PP_REQUIRE(img_is_f32 || (mean_host && std_host), PP_ERR_INVALID_ARG,
               "pp_preproc_im2col: mean/std are required for uint8 input");
    const float zeros[3] = {0.f, 0.f, 0.f}, ones[3] = {1.f, 1.f, 1.f};
    if (img_is_f32) {
        mean_host = zeros;
        std_host = ones;
    }
    PP_REQUIRE(patch == 16, PP_ERR_UNSUPPORTED, "pp_preproc_im2col: patch size must be 16");
    PP_REQUIRE(passes == 1 || passes == 2, PP_ERR_INVALID_ARG, "pp_preproc_im2col: passes must be 1 or 2");
    PP_REQUIRE(B > 0 && H > 0 && W > 0 && pad >= 0, PP_ERR_INVALID_ARG, "pp_preproc_im2col: bad shape");
    const int Hp = (H + 2 * pad - patch) / patch + 1, Wp = (W + 2 * pad - patch) / patch + 1;
    const long long total = (long long)passes * B * Hp * Wp * 3 * patch;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PP_IM2COL(T, TIn)                                                                                         \
    hipLaunchKernelGGL((preproc_im2col_kernel<T, TIn>), grid, block, 0, s, reinterpret_cast<const TIn*>(img),       \
                       reinterpret_cast<T*>(patches), B, passes, H, W, Hp, Wp, pad, mean_host[0], mean_host[1],     \
                       mean_host[2], std_host[0], std_host[1], std_host[2], bgr_to_rgb)
    if (prec == PP_PREC_BF16) {
        if (img_is_f32) PP_IM2COL(__bf16, float); else PP_IM2COL(__bf16, uint8_t);
    } else if (prec == PP_PREC_F32) {
        if (img_is_f32) PP_IM2COL(float, float); else PP_IM2COL(float, uint8_t);
    } else if (prec == PP_PREC_F16X3) {
        if (img_is_f32) PP_IM2COL(SplitH, float); else PP_IM2COL(SplitH, uint8_t);
    } else
        return fail(PP_ERR_INVALID_ARG, "pp_preproc_im2col: unknown precision");
#undef PP_IM2COL
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_layernorm(const float* x, const float* gamma, const float* beta, void* y, int M, int E, float eps,
                            int out_bf16, void* stream) {
    using namespace pp;
    PP_REQUIRE(x && gamma && beta && y, PP_ERR_INVALID_ARG, "pp_layernorm: NULL argument");
    PP_REQUIRE(M > 0, PP_ERR_INVALID_ARG, "pp_layernorm: M must be positive");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (out_bf16 == PP_OUT_SPLIT) return launch_ln<SplitH>(x, gamma, beta, y, M, E, eps, s);
    return out_bf16 ? launch_ln<__bf16>(x, gamma, beta, y, M, E, eps, s) : launch_ln<float>(x, gamma, beta, y, M, E, eps, s);
}
