/*
 * probpose_mi355x.h -- C ABI of libprobpose_mi355x.so
 *
 * MI355X (gfx950 / CDNA4) implementation of the ProbPose top-down inference hot path:
 * ViT backbone forward -> ProbMapHead (deconv heatmap branch + Sparsemax, four scalar
 * towers, flip-test averaging) -> ProbMap decode (OKS-kernel convolution, argmax,
 * sub-pixel step). Each entry point replaces one stage of the reference's Python path;
 * the reference line range it stands in for is cited above the declaration (paths are
 * relative to the reference tree, MiraPurkrabek/ProbPose_code).
 *
 * Conventions:
 *   - plain C: raw device pointers + explicit sizes, no framework types;
 *   - every function returns an int status: PP_OK (0) or a negative PP_ERR_* code;
 *     pp_last_error() gives the thread-local message of the last failure;
 *   - the library never allocates or frees device memory, never synchronises the
 *     stream and never throws; outputs and workspaces are caller-allocated;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     enqueued on it and the call returns immediately;
 *   - all pointers are DEVICE pointers unless the parameter is named *_host.
 */
#ifndef PROBPOSE_MI355X_H_
#define PROBPOSE_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 1

enum {
    PP_OK = 0,
    PP_ERR_INVALID_ARG = -1, /* NULL pointer, non-positive size, inconsistent shapes        */
    PP_ERR_UNSUPPORTED = -2, /* shape / option outside what the kernels were built for      */
    PP_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString)   */
    PP_ERR_WORKSPACE = -4    /* caller-provided workspace too small                          */
};

/* Largest OKS-kernel radius the decode kernels support (reference clips the kernel
 * variance to 3.0 => radius ceil(3*3.0) = 9, post_processing.py:23-24). */
#define PP_MAX_RADIUS 9
#define PP_MAX_TAPS (2 * PP_MAX_RADIUS + 1)

int pp_abi_version(void);
const char* pp_last_error(void);
const char* pp_status_string(int status);

/* Number of compute units of the current device (for callers sizing batches); <0 on error. */
int pp_device_cu_count(void);

/* ------------------------------------------------------------------------------------
 * ProbMap decode, fused with the flip-test average.
 *
 * Replaces, per batch, the reference's per-sample Python loop
 *   flip_heatmaps(...)            mmpose/models/utils/tta.py:35-39
 *   (_htm + _htm_flip) * 0.5      mmpose/models/heads/hybrid_heads/probmap_head.py:763
 *   BaseHead.decode               mmpose/models/heads/base_head.py:33-86
 *   ProbMap.decode                mmpose/codecs/probmap.py:170-220
 *   get_heatmap_expected_value    mmpose/codecs/utils/post_processing.py:308-381
 *   _get_subpixel_maximums        mmpose/codecs/utils/post_processing.py:384-430
 *
 * hm          (B,K,H,W) f32 probability maps of the un-flipped pass.
 * hm_flip     (B,K,H,W) f32 maps of the horizontally flipped pass, or NULL (no flip test).
 *             When given, the map decoded for (b,k) is
 *             (hm[b,k,y,x] + hm_flip[b,flip_indices[k],y,W-1-x]) * 0.5f.
 * flip_indices (K) i32; required iff hm_flip != NULL.
 * taps        (K, PP_MAX_TAPS) f64: row k holds the 2*radius[k]+1 normalised 1-D factors
 *             of keypoint k's OKS kernel (the 2-D kernel of post_processing.py:13-39 is the
 *             outer product of this vector with itself up to f64 rounding), rest ignored.
 * radius      (K) i32, each in [0, PP_MAX_RADIUS].
 * in_w, in_h  codec input_size; keypoints = locs / (W-1, H-1) * (in_w, in_h) in f64.
 * avg_out     optional (B,K,H,W) f32: the averaged map that was decoded (pred_fields.heatmaps).
 * conv_out    optional (B,K,H,W) f32: the OKS-convolved map (f64 accumulate, rounded once).
 * locs        (B,K,2) f32 heatmap-space (x,y) after the sub-pixel step.
 * keypoints   (B,K,2) f64 input-pixel space (what ProbMap.decode returns, probmap.py:218).
 * scores      (B,K) f32: the un-convolved map at the integer argmax (keypoints_conf).
 * ---------------------------------------------------------------------------------- */
int pp_probmap_decode(const float* hm, const float* hm_flip, const int32_t* flip_indices,
                      const double* taps, const int32_t* radius,
                      int B, int K, int H, int W, double in_w, double in_h,
                      float* avg_out, float* conv_out,
                      float* locs, double* keypoints, float* scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROBPOSE_MI355X_H_ */
