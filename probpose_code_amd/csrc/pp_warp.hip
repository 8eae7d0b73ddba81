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

}  // namespace pp

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
