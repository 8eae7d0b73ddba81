// Top-down crop extraction on the device: for each person box one cv2.warpAffine(img, M, (W, H), INTER_LINEAR) with
// constant-zero border (TopdownAffine.transform, mmpose/datasets/transforms/topdown_transforms.py:118-126) written
// straight in the CHW uint8 layout PackPoseInputs / image_to_tensor produce (mmpose/datasets/transforms/formatting.py:14-36).
//
// The arithmetic is OpenCV's fixed-point bilinear warp restated from its published source (imgproc/imgwarp.cpp,
// WarpAffineInvoker + remapBilinear, opencv 4.x - a third-party dependency of the reference, un-vendored, cv2 is absent
// in this image: PARITY UNPINNED):
//   * the caller passes the INVERSE map (dst -> src) in float64, inverted the way warpAffine does;
//   * X = (round((M1 y + M2) 1024) + 16 + round(M0 x 1024)) >> 5  - coordinates with 5 fractional bits;
//   * the four taps are weighted with 15-bit integer weights (32 - fx)(32 - fy) 32, ..., taps outside the image read
//     the border value 0, result = (sum + 16384) >> 15.
// One thread per output pixel (all three channels): HBM-bound on the 3 x 48 KiB each crop writes.
#include "pp_common.h"

#include <cstdint>

namespace pp {

__device__ __forceinline__ int round_to_int(double v) {  // saturate_cast<int>(double) = cvRound: nearest, ties to even
    v = rint(v);
    return v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (int)0x80000000 : (int)v);
}

__global__ __launch_bounds__(256) void warp_affine_kernel(const uint8_t* __restrict__ img, int ih, int iw, int ic,
                                                          const double* __restrict__ inv, uint8_t* __restrict__ out,
                                                          int n, int oh, int ow) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= ow) return;
    const double* M = inv + 6 * b;
    const int X0 = round_to_int((M[1] * y + M[2]) * 1024.0) + 16, Y0 = round_to_int((M[4] * y + M[5]) * 1024.0) + 16;
    const int X = (X0 + round_to_int(M[0] * x * 1024.0)) >> 5, Y = (Y0 + round_to_int(M[3] * x * 1024.0)) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);  // saturate_cast<short>
    sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0in = sx >= 0 && sx < iw, x1in = sx + 1 >= 0 && sx + 1 < iw;
    const bool y0in = sy >= 0 && sy < ih, y1in = sy + 1 >= 0 && sy + 1 < ih;
    for (int c = 0; c < ic; ++c) {
        const uint8_t* s = img + ((size_t)sy * iw + sx) * ic + c;
        const int v00 = (x0in && y0in) ? s[0] : 0, v01 = (x1in && y0in) ? s[ic] : 0;
        const int v10 = (x0in && y1in) ? s[(size_t)iw * ic] : 0, v11 = (x1in && y1in) ? s[(size_t)iw * ic + ic] : 0;
        const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        out[(((size_t)b * ic + c) * oh + y) * ow + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// Heatmaps of the persons of one image back on the image (revert_heatmap, mmpose/structures/utils.py:146-175: float32
// cv2.warpAffine(heatmap, M_n, (W, H), INTER_LINEAR), zero border) merged by the element-wise maximum over the persons
// (merge_data_samples, :121-123) without materialising the N full-size maps: one thread per image pixel, N x K taps from
// the (L2-resident) person maps, K running maxima in registers, one coalesced store per channel. HBM-bound on the
// K x H x W floats written. cv2's float path: the same 5-fractional-bit source coordinates as above, the four weights
// (1 - fy/32)(1 - fx/32), ... in float32 (exact), v = s00 w00 + s01 w01 + s10 w10 + s11 w11.
constexpr int RV_MAXK = 32;

__global__ __launch_bounds__(256) void revert_heatmaps_max_kernel(const float* __restrict__ hm, const double* __restrict__ inv,
                                                                  float* __restrict__ out, int n, int K, int hh, int hw,
                                                                  int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float acc[RV_MAXK];
#pragma unroll
    for (int k = 0; k < RV_MAXK; ++k) acc[k] = -3.402823466e+38f;
    for (int b = 0; b < n; ++b) {
        const double* M = inv + 6 * b;
        const int X0 = round_to_int((M[1] * y + M[2]) * 1024.0) + 16, Y0 = round_to_int((M[4] * y + M[5]) * 1024.0) + 16;
        const int X = (X0 + round_to_int(M[0] * x * 1024.0)) >> 5, Y = (Y0 + round_to_int(M[3] * x * 1024.0)) >> 5;
        int sx = X >> 5, sy = Y >> 5;
        sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
        sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
        const bool x0in = sx >= 0 && sx < hw, x1in = sx + 1 >= 0 && sx + 1 < hw;
        const bool y0in = sy >= 0 && sy < hh, y1in = sy + 1 >= 0 && sy + 1 < hh;
        if (!((x0in || x1in) && (y0in || y1in))) {  // this person's map is 0 here (border value)
#pragma unroll
            for (int k = 0; k < RV_MAXK; ++k) acc[k] = fmaxf(acc[k], 0.f);
            continue;
        }
        const float ax = (float)(X & 31) * (1.f / 32.f), ay = (float)(Y & 31) * (1.f / 32.f);
        const float w00 = (1.f - ay) * (1.f - ax), w01 = (1.f - ay) * ax, w10 = ay * (1.f - ax), w11 = ay * ax;
        const float* s = hm + (size_t)b * K * hh * hw + (ptrdiff_t)sy * hw + sx;
#pragma unroll
        for (int k = 0; k < RV_MAXK; ++k) {
            if (k < K) {
                const float* sk = s + (size_t)k * hh * hw;
                const float v00 = (x0in && y0in) ? sk[0] : 0.f, v01 = (x1in && y0in) ? sk[1] : 0.f;
                const float v10 = (x0in && y1in) ? sk[hw] : 0.f, v11 = (x1in && y1in) ? sk[hw + 1] : 0.f;
                const float v = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
                acc[k] = fmaxf(acc[k], v);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < RV_MAXK; ++k)
        if (k < K) out[((size_t)k * H + y) * W + x] = acc[k];
}

// heatmaps / heatmaps.sum(axis=(1, 2)) * presence[k]   (local_visualizer.py:827-837), two launches: per-channel partial sums
// in float64 (fixed order), then the scaling
constexpr int PS_PARTS = 64;
__global__ __launch_bounds__(256) void channel_partial_sum_kernel(const float* __restrict__ hm, double* __restrict__ parts, int HW) {
    __shared__ double red[256];
    const int k = blockIdx.y, part = blockIdx.x;
    const size_t per = ((size_t)HW + PS_PARTS - 1) / PS_PARTS, lo = part * per, hi = lo + per < (size_t)HW ? lo + per : (size_t)HW;
    double s = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)hm[(size_t)k * HW + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[k * PS_PARTS + part] = red[0];
}

__global__ __launch_bounds__(256) void channel_scale_kernel(float* __restrict__ hm, const double* __restrict__ parts,
                                                            const float* __restrict__ presence, int HW) {
    const int k = blockIdx.y;
    double s = 0.0;
    for (int i = 0; i < PS_PARTS; ++i) s += parts[k * PS_PARTS + i];
    const float total = (float)s, pr = presence[k];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)HW) hm[(size_t)k * HW + i] = hm[(size_t)k * HW + i] / total * pr;
}

}  // namespace pp

extern "C" int pp_revert_heatmaps_max(const float* heatmaps, const double* inverse_maps, float* out, int n, int K, int hm_h,
                                      int hm_w, int img_h, int img_w, void* stream) {
    using namespace pp;
    PP_REQUIRE(heatmaps && inverse_maps && out, PP_ERR_INVALID_ARG, "pp_revert_heatmaps_max: NULL argument");
    PP_REQUIRE(n > 0 && K > 0 && K <= RV_MAXK && hm_h > 0 && hm_w > 0 && img_h > 0 && img_w > 0, PP_ERR_INVALID_ARG,
               "pp_revert_heatmaps_max: bad shape (at most 32 channels)");
    PP_REQUIRE(hm_h < 32768 && hm_w < 32768 && img_h <= 65535, PP_ERR_UNSUPPORTED, "pp_revert_heatmaps_max: sides too large");
    hipLaunchKernelGGL(revert_heatmaps_max_kernel, dim3((img_w + 255) / 256, img_h), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), heatmaps, inverse_maps, out, n, K, hm_h, hm_w, img_h, img_w);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_heatmap_posterior(float* heatmaps, const float* presence, double* scratch, int K, int H, int W, void* stream) {
    using namespace pp;
    PP_REQUIRE(heatmaps && presence && scratch, PP_ERR_INVALID_ARG, "pp_heatmap_posterior: NULL argument");
    PP_REQUIRE(K > 0 && H > 0 && W > 0 && K <= 65535, PP_ERR_INVALID_ARG, "pp_heatmap_posterior: bad shape");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(channel_partial_sum_kernel, dim3(PS_PARTS, K), dim3(256), 0, s, heatmaps, scratch, H * W);
    PP_LAUNCH_CHECK();
    hipLaunchKernelGGL(channel_scale_kernel, dim3((H * W + 255) / 256, K), dim3(256), 0, s, heatmaps, scratch, presence, H * W);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_warp_affine_u8(const void* img_hwc, int img_h, int img_w, int channels, const double* inverse_maps,
                                 void* crops_chw, int n, int out_h, int out_w, void* stream) {
    using namespace pp;
    if (n == 0) return PP_OK;
    PP_REQUIRE(img_hwc && inverse_maps && crops_chw, PP_ERR_INVALID_ARG, "pp_warp_affine_u8: NULL argument");
    PP_REQUIRE(n > 0 && img_h > 0 && img_w > 0 && out_h > 0 && out_w > 0 && channels > 0 && channels <= 4, PP_ERR_INVALID_ARG,
               "pp_warp_affine_u8: bad shape");
    PP_REQUIRE(img_h < 32768 && img_w < 32768 && out_h <= 65535 && n <= 65535, PP_ERR_UNSUPPORTED,
               "pp_warp_affine_u8: image sides must be below 32768 (16-bit source coordinates, as in cv2)");
    hipLaunchKernelGGL(warp_affine_kernel, dim3((out_w + 255) / 256, out_h, n), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint8_t*>(img_hwc), img_h, img_w, channels,
                       inverse_maps, reinterpret_cast<uint8_t*>(crops_chw), n, out_h, out_w);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
