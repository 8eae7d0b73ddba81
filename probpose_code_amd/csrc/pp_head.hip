// Small memory-bound pieces of ProbMapHead's four scalar towers (probability / visibility /
// oks / error; mmpose/models/heads/hybrid_heads/probmap_head.py:261-410) for gfx950.
// The 3x3 convolutions (with BatchNorm folded in) run in pp_gemm.hip as implicit GEMMs over
// NHWC activations; what is left is MaxPool + ReLU between them and the final
// Conv1x1 -> Sigmoid/ReLU -> flip-test average on the 1x1 feature.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// four consecutive elements at element index idx (idx % 4 == 0) of a tensor in one of the three activation formats
__device__ __forceinline__ f32x4 load4(const float* base, size_t idx) { return *reinterpret_cast<const f32x4*>(base + idx); }
__device__ __forceinline__ f32x4 load4(const __bf16* base, size_t idx) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(base + idx);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 load4(const SplitH* base, size_t idx) { return split_load4(base, idx); }
__device__ __forceinline__ void store4(float* base, size_t idx, f32x4 v) { *reinterpret_cast<f32x4*>(base + idx) = v; }
__device__ __forceinline__ void store4(__bf16* base, size_t idx, f32x4 v) {
    *reinterpret_cast<bf16x4*>(base + idx) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
__device__ __forceinline__ void store4(SplitH* base, size_t idx, f32x4 v) { split_store4(base, idx, v); }

// MaxPool2d(kernel = stride = (ph, pw), no padding, floor) followed by ReLU on NHWC tensors.
// in  [N, H, W, C] -> out [N, H/ph, W/pw, C]; one thread per 4 output channels.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void maxpool_relu_kernel(const TI* __restrict__ in, TO* __restrict__ out, int N,
                                                           int H, int W, int C, int ph, int pw) {
    const int Ho = H / ph, Wo = W / pw, C4 = C / 4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * Ho * Wo * C4) return;
    const int c = (int)(gid % C4) * 4;
    long long r = gid / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int n = (int)(r / Ho);
    f32x4 m = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
            const f32x4 v = load4(in, (((size_t)n * H + yo * ph + i) * W + xo * pw + j) * C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], 0.f);
    store4(out, (((size_t)n * Ho + yo) * Wo + xo) * C + c, m);
}

// The same pooling on split-K partial sums: in = sum over `nsplit` fp32 slices (slice stride `split_stride` elements)
// + bias[n / images_per_group][c]; the 3x3 convolutions of the later tower stages have so few output rows that one
// workgroup per output tile leaves most of the chip idle, so their K loop is cut in three (pp_conv3x3_splitk) and the
// reduction is folded in here.
// NS > 0: the slice count at compile time - a position's NS loads are then independent and in flight together (with the runtime loop the compiler
// chains them: 108 dependent L2 round trips per thread on a (4, 3) window of 9 slices, 18 - 22 us for a launch of 48 workgroups)
template <typename TO, int NS = 0>
__global__ __launch_bounds__(256) void sum_maxpool_relu_kernel(const float* __restrict__ in, int nsplit, long long split_stride,
                                                               const float* __restrict__ bias, int images_per_group,
                                                               TO* __restrict__ out, int N, int H, int W, int C, int ph,
                                                               int pw) {
    const int Ho = H / ph, Wo = W / pw, C4 = C / 4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * Ho * Wo * C4) return;
    const int c = (int)(gid % C4) * 4;
    long long r = gid / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int n = (int)(r / Ho);
    f32x4 m = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
            const float* src = in + (((size_t)n * H + yo * ph + i) * W + xo * pw + j) * C + c;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            if constexpr (NS > 0) {
                f32x4 part[NS];
#pragma unroll
                for (int sp = 1; sp < NS; ++sp) part[sp] = *reinterpret_cast<const f32x4*>(src + sp * split_stride);
#pragma unroll
                for (int sp = 1; sp < NS; ++sp) v += part[sp];  // (the same order as the runtime loop: same bits)
            } else {
                for (int sp = 1; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4*>(src + sp * split_stride);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
        }
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + (size_t)(n / images_per_group) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q] + b[q], 0.f);  // max(x) + b == max(x + b): the bias is per channel
    store4(out, (((size_t)n * Ho + yo) * Wo + xo) * C + c, m);
}

// Final layer of the four towers + flip-test average (probmap_head.py:766-774).
//   feat [4][passes*B][C] (1x1 spatial), w [4][K][C] fp32, bias [4][K] fp32
//   out  [4][B][K] fp32:  act(w.f + b) for the un-flipped crop, averaged with the flipped
//   crop's channel flip_indices[k]; act = sigmoid for towers 0..2, ReLU for tower 3 (error);
//   tower 3 is additionally divided by err_div (sqrt(H^2 + W^2) of the heatmap, :786-787).
// One wavefront per (tower, crop, keypoint).
template <typename TI>
__global__ __launch_bounds__(256) void tower_final_kernel(const TI* __restrict__ feat, const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const int32_t* __restrict__ flip_indices,
                                                          float* __restrict__ out, int B, int passes, int C, int K,
                                                          float err_div) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= 4 * B * K) return;
    const int k = wid % K, b = (wid / K) % B, t = wid / (K * B);
    float res = 0.f;
    for (int pass = 0; pass < passes; ++pass) {
        const int kk = pass ? flip_indices[k] : k;
        const size_t f = ((size_t)t * passes * B + (size_t)pass * B + b) * C;
        const float* wr = w + ((size_t)t * K + kk) * C;
        float acc = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 fv = load4(feat, f + c), wv = load4(wr, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_fmaf(fv[q], wv[q], acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        acc += bias[t * K + kk];
        res += (t == 3) ? fmaxf(acc, 0.f) : 1.0f / (1.0f + expf(-acc));
    }
    if (passes == 2) res *= 0.5f;
    if (t == 3) res = res / err_div;
    if (lane == 0) out[wid] = res;
}

}  // namespace pp

extern "C" int pp_maxpool_relu_nhwc(const void* in, int in_bf16, void* out, int out_bf16, int N, int H, int W, int C,
                                    int ph, int pw, void* stream) {
    using namespace pp;
    PP_REQUIRE(in && out, PP_ERR_INVALID_ARG, "pp_maxpool_relu_nhwc: NULL argument");
    PP_REQUIRE(N > 0 && C % 4 == 0 && ph > 0 && pw > 0 && H >= ph && W >= pw, PP_ERR_INVALID_ARG,
               "pp_maxpool_relu_nhwc: bad shape (C must be a multiple of 4)");
    const long long total = (long long)N * (H / ph) * (W / pw) * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PP_MP(TI, TO)                                                                                             \
    hipLaunchKernelGGL((maxpool_relu_kernel<TI, TO>), grid, block, 0, s, reinterpret_cast<const TI*>(in),         \
                       reinterpret_cast<TO*>(out), N, H, W, C, ph, pw)
    PP_REQUIRE(in_bf16 >= 0 && in_bf16 <= 2 && out_bf16 >= 0 && out_bf16 <= 2 && (in_bf16 != 2 || out_bf16 != 1) &&
                   (in_bf16 != 1 || out_bf16 != 2) && ((in_bf16 != 2 && out_bf16 != 2) || C % 32 == 0),
               PP_ERR_UNSUPPORTED, "pp_maxpool_relu_nhwc: formats are PP_OUT_F32 / _BF16 / _SPLIT (bf16 and split do not mix; split needs C % 32 == 0)");
    if (in_bf16 == 2 && out_bf16 == 2) PP_MP(SplitH, SplitH);
    else if (in_bf16 == 2) PP_MP(SplitH, float);
    else if (out_bf16 == 2) PP_MP(float, SplitH);
    else if (in_bf16 && out_bf16) PP_MP(__bf16, __bf16);
    else if (in_bf16) PP_MP(__bf16, float);
    else if (out_bf16) PP_MP(float, __bf16);
    else PP_MP(float, float);
#undef PP_MP
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_tower_final(const void* feat, int feat_bf16, const float* w, const float* bias,
                              const int32_t* flip_indices, float* out, int B, int passes, int C, int K, float err_div,
                              void* stream) {
    using namespace pp;
    PP_REQUIRE(feat && w && bias && out, PP_ERR_INVALID_ARG, "pp_tower_final: NULL argument");
    PP_REQUIRE(passes == 1 || (passes == 2 && flip_indices), PP_ERR_INVALID_ARG,
               "pp_tower_final: passes must be 1, or 2 with flip_indices");
    PP_REQUIRE(B > 0 && K > 0 && C % 4 == 0, PP_ERR_INVALID_ARG, "pp_tower_final: bad shape");
    const dim3 grid((4 * B * K + 3) / 4), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (feat_bf16 == 2) {
        PP_REQUIRE(C % 32 == 0, PP_ERR_UNSUPPORTED, "pp_tower_final: split-fp16 features need C % 32 == 0");
        hipLaunchKernelGGL(tower_final_kernel<SplitH>, grid, block, 0, s, reinterpret_cast<const SplitH*>(feat), w, bias,
                           flip_indices, out, B, passes, C, K, err_div);
    } else if (feat_bf16)
        hipLaunchKernelGGL(tower_final_kernel<__bf16>, grid, block, 0, s, reinterpret_cast<const __bf16*>(feat), w, bias,
                           flip_indices, out, B, passes, C, K, err_div);
    else
        hipLaunchKernelGGL(tower_final_kernel<float>, grid, block, 0, s, reinterpret_cast<const float*>(feat), w, bias,
                           flip_indices, out, B, passes, C, K, err_div);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_sum_maxpool_relu_nhwc(const float* partials, int nsplit, long long split_stride, const float* bias,
                                        int images_per_group, void* out, int out_bf16, int N, int H, int W, int C, int ph,
                                        int pw, void* stream) {
    using namespace pp;
    PP_REQUIRE(partials && out, PP_ERR_INVALID_ARG, "pp_sum_maxpool_relu_nhwc: NULL argument");
    PP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ph > 0 && pw > 0 && H >= ph && W >= pw && nsplit >= 1 &&
                   images_per_group >= 1,
               PP_ERR_INVALID_ARG, "pp_sum_maxpool_relu_nhwc: bad shape");
    const long long total = (long long)N * (H / ph) * (W / pw) * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (out_bf16 == 2) {
        PP_REQUIRE(C % 32 == 0, PP_ERR_UNSUPPORTED, "pp_sum_maxpool_relu_nhwc: split-fp16 output needs C % 32 == 0");
        auto kern = nsplit == 9 ? sum_maxpool_relu_kernel<SplitH, 9> : nsplit == 4 ? sum_maxpool_relu_kernel<SplitH, 4>
                    : nsplit == 3 ? sum_maxpool_relu_kernel<SplitH, 3> : sum_maxpool_relu_kernel<SplitH, 0>;
        hipLaunchKernelGGL(kern, grid, block, 0, s, partials, nsplit, split_stride, bias, images_per_group, reinterpret_cast<SplitH*>(out), N, H, W, C, ph,
                           pw);
    } else if (out_bf16)
        hipLaunchKernelGGL(sum_maxpool_relu_kernel<__bf16>, grid, block, 0, s, partials, nsplit, split_stride, bias,
                           images_per_group, reinterpret_cast<__bf16*>(out), N, H, W, C, ph, pw);
    else
        hipLaunchKernelGGL(sum_maxpool_relu_kernel<float>, grid, block, 0, s, partials, nsplit, split_stride, bias,
                           images_per_group, reinterpret_cast<float*>(out), N, H, W, C, ph, pw);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// The fixed-layout result record of the multi-GPU exchange (probpose_code_amd/dist.py): per (crop, keypoint) seven float64
// [x, y, conf, prob, vis, oks, err] from the decode outputs (keypoints f64, scores f32) and the four tower scalars
// (tower-major f32) - what collect_results carries as pickled dicts in the reference.
namespace pp {
__global__ void pack_records_kernel(const double* __restrict__ kpts, const float* __restrict__ scores,
                                    const float* __restrict__ scalars, double* __restrict__ rec, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* r = rec + (size_t)i * 7;
    r[0] = kpts[2 * i];
    r[1] = kpts[2 * i + 1];
    r[2] = (double)scores[i];
#pragma unroll
    for (int t = 0; t < 4; ++t) r[3 + t] = (double)scalars[(size_t)t * n + i];
}
}  // namespace pp

extern "C" int pp_pack_records(const double* keypoints, const float* scores, const float* scalars, double* records, int n,
                               void* stream) {
    using namespace pp;
    if (n == 0) return PP_OK;
    PP_REQUIRE(keypoints && scores && scalars && records && n > 0, PP_ERR_INVALID_ARG, "pp_pack_records: NULL argument or negative count");
    hipLaunchKernelGGL(pack_records_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keypoints,
                       scores, scalars, records, n);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
