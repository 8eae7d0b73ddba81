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

/* 4 (round 6): the numeric domain of PP_PREC_F16X3 stated and guarded (below); power-of-two WEIGHT SCALES: + pp_gemm_ws, pp_linear_ln_folded_ws,
 *    pp_qkv_attention_split_ws, pp_gemm_residual_layernorm_ws, pp_ffn_split_residual_layernorm_ws, pp_proj_ffn_split_residual_layernorm_ws (the unsuffixed entry points = scale 1);
 *    CHANGED signatures: pp_qkv_attention_split_folded (centered rows, no column sums, + w_inv_scale), pp_proj_ffn_split_folded (+ residual_stats,
 *    + three weight scales; the rows it leaves are centered); pp_probmap_decode_flags writes NaN results for a map with a non-finite logit;
 *    + pp_skinny_linear / pp_skinny_linear_tile / pp_skinny_deconv / pp_skinny_conv1x1_planar (the launch plan of small batches); REMOVED: the eight-wave feed-forward kernel, the overlapped-epilogue Linear kernel and the
 *    head-pair qkv + attention kernel with their options "ffn_dma_waves" / "linear_ovl" / "qkv_attn_pair".
 * 3: + pp_launch_count / pp_reset_launch_counts (diagnostics: which kernels a launch plan really ran); pp_linear_ln_folded, the *_folded launches
 *    and PP_WS_LN_STATS (round 5).
 * 2: + pp_clock_probe, pp_conv3x3_splitk_slices, pp_conv3x3_winograd_maxpool_relu, pp_winograd_scratch_bytes, pp_probmap_decode_flags, option "ksplit9_below"; PP_WS_TOWER_PARTIAL sized for the slice count the
 *    library itself picks (round 3 added pp_workspace_bytes / pp_set_option / the split-fp16 layer kernels under version 1). */
#define PP_ABI_VERSION 4

/* NUMERIC DOMAIN of PP_PREC_F16X3 (the parity mode: every parity figure of the repo is taken in it).
 * An operand x is carried as hi = fp16(x), lo = fp16(x - hi) (csrc/pp_split.h); a product is three fp16 MFMAs with fp32 accumulation.
 *   range      |x| <= 65504 for EVERY MFMA operand: activations (LayerNorm outputs, q / k / v, the FFN's hidden activation, features, deconvolution
 *              maps) and - since the LayerNorm fold - the residual rows themselves (centered per row in the ViT-S chain, raw in ViT-B's
 *              pp_linear_ln_folded plan). Beyond it hi = inf, the product NaN: it reaches the heatmap logits as NaN / inf, pp_probmap_decode_flags
 *              then writes NaN keypoints / scores (never "pixel 0 with a score"), and the host mirror raises FloatingPointError.
 *   precision  hi + lo has 22 significant bits while lo is a NORMAL fp16 number: |x| >= 2^-3. Below that lo is subnormal and the pair is exact
 *              to an ABSOLUTE 2^-25 only (no bits at all below 2^-25). Harmless for activations next to O(1) neighbours, fatal for weights,
 *              which are small numbers throughout (trained ViT weights ~ 0.02: 17 bits; rows of 1e-3: 13 bits - measured, tests/
 *              test_trained_stats.py). Linear weights are therefore stored SCALED by a power of two, the tensor's largest element in
 *              [2^12, 2^13) (probpose_code_amd/weights.py), and the kernels multiply their accumulators by the inverse (the *_ws entry points,
 *              `w_inv_scale`: exact). Elements down to 2^-16 of a tensor's largest keep all 22 bits.
 *   weights    a weight (after the BatchNorm / LayerNorm-gamma folds) with |w| > 65504 or non-finite is refused at load (weights.check_split_range).
 *   LayerNorm fold   ViT-S chain: rows travel CENTERED, the folded projection is rstd * ((x - mean) W'^T) + b' - the accuracy of a plain LayerNorm.
 *              ViT-B plan (pp_linear_ln_folded): rstd * (x W'^T - mean * colsum) + b' on raw rows loses ~ log2(1 + |mean| / std) bits of the fp32
 *              accumulator: 2 - 5e-6 relative at |mean| / std <= 1 (a trained ViT's rows), 3e-5 at 10, 5e-4 at 150; engine plan ln_fold=False
 *              applies the LayerNorm before the split instead.
 *   diagnosis  probpose_code_amd.domain_report(state_dict, crops, heads) runs a checkpoint once in fp32 and reports every operand's range and the
 *              rows' |mean| / std per layer.
 * PP_PREC_F32 has fp32's range and no such limits (16x slower MFMA); PP_PREC_BF16 has fp32's range and 8 bits. */

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

/* A HIP stream of the caller's own (non-blocking), and its end. For hosts that fork work to a second stream INSIDE a stream capture: on ROCm 7.0 a
 * stream that took part in a capture whose graph has since been destroyed must not be forked to in a later capture (the first launch of the new
 * graph crashes inside hipGraphLaunch) - a stream per captured graph, destroyed with it, is the safe pattern (probpose_code_amd/engine.py, capture). */
int pp_stream_create(void** stream_out);
int pp_stream_destroy(void* stream);

/* Explicit process-wide options. The library NEVER reads the environment: kernel selection depends only on the arguments of a
 * call and on these switches, which exist for A/B timing and default to the shipped plan:
 *   "panel" (1)              0: every GEMM / convolution on the 128 x 128-tile kernel
 *   "conv_halo" (1)          0: first tower convolution (bf16) on the implicit-GEMM kernel
 *   "psplit_nst" (0)         2 / 3: force the two- / three-stage form of the wide-tile split kernel
 *   "panel_linear_mink" (0)  > 0: shortest K of a bf16 Linear layer that takes the wide-tile kernel
 *   "psplit_bf16_conv" (0)   1: bf16 convolutions through the split kernel's bf16 instantiation
 *   "psplit_tap_inner" (1)          K walk of the gathered split-fp16 convolutions: 1 = 3x3 convolutions channel-block-major
 *                                   (the nine shifted re-reads of a block hit L2), 2 = deconvolutions too, 0 = tap-major
 *   "psplit_conv_weight_major" (1)  tiles of those 3x3 convolutions weight-set-major (with taps inner only): half the HBM-side fetch
 *   "attn_dma" (1)           0: split-fp16 attention of 432-token sequences with the register-staged kernel
 *   "conv_pool_split" (1)    0: split-fp16 first tower stage as two launches (conv, then pooling)
 *   "decode_wgs_per_cu" (3)  pp_probmap_(head_)decode: workgroups per CU its LDS band buffer is sized for (5 .. 1)
 *   "ksplit9_below" (1024)   pp_conv3x3_splitk_slices: output rows under which a small tower stage is cut into nine K-slices
 *   "linear_dma" (1)         0: large split-fp16 Linear layers (pp_gemm, N % 192 == 0, >= 512 tiles) stay on the wide-tile kernel instead of the
 *                            twelve-wave 192 x 192 kernel (pp_linear_dma.hip: eight computing + four DMA-only waves)
 *   "linear_loop" (1)        twelve-wave Linear kernel: one workgroup per CU walks a column of tiles, the next tile's first stages requested
 *                            under this tile's epilogue; 0: a workgroup per tile
 *   "psplit_deconv_weight_major" (0)  dev A/B: 1 = deconvolution tiles weight-set(phase)-major per XCD instead of row panel -> phase
 *   "ffn_pair" (1)           twelve-wave feed-forward launch: hidden chunks in pairs that share every streamed x k-block (x rows streamed 6 instead of 12
 *                            times per launch; taken when F / 128 is even); 0: one chunk at a time
 *   "psplit_tail" (1)        0: split-fp16 Linear layers never send the rows of a ragged last round to a second launch on 128 x 192 tiles
 *   "wino_order" (8)         pp_conv3x3_winograd_maxpool_relu: column tiles per 32-workgroup super tile (0: column tiles fastest)
 *   "qkv_attn_deep" (1)      pp_qkv_attention_split(_ws) of a small launch (<= 2 workgroups per CU): ring of four stages, one workgroup per CU (0: always two stages)
 *   "qkv_attn_qsplit" (1)    ... and while 2 x sequences x heads <= 0.8 x CUs: two workgroups per (sequence, head), each attends for half of the query tiles
 *   "skinny_tile" (0)        pp_skinny_linear: 10 RT + CT (11, 22, 33, 13, 12, 23) forces a tile of 32 RT rows x 32 CT columns (0: the cost rule of pp_skinny.hip)
 *   "skinny_xcd_order" (1)   pp_skinny_linear launches with a LayerNorm tail and >= 2 048 rows: tiles ordered so that an XCD touches 1 / xr of the row
 *                            blocks and 1 / xc of the column tiles (xr xc = 8; 0: row-major)
 *   "ksplit_channels" (1)    pp_conv3x3_splitk_slices: 0 = whole-tap slices only (never the four channel ranges of the wide-tile kernel)
 * Unknown names return PP_ERR_INVALID_ARG. Not thread-safe against concurrent launches (set them before the first call).
 * (probpose_code_amd/_lib.py forwards PP_OPT_<NAME>=<int> environment variables here at import - host-side convenience.) */
int pp_set_option(const char* name, int value);
int pp_get_option(const char* name, int* value);

/* Diagnostics: kernel launches since the last reset, tallied on the host at launch time (a captured hipGraph counts once, at capture) under
 * the launching source file's name - "pp_winograd.hip", "pp_ffn_dma.hip", "pp_qkv_attn_split.hip", "pp_linear_dma.hip", "pp_gemm.hip" ... - and
 * for kernels that share a file under their own tag: "linear_dma_tile" (twelve-wave Linear kernel under pp_gemm), "linear_dma_fold" (the same
 * kernel under pp_linear_ln_folded), "ffn_dma_pair" / "ffn_dma_single" / "ffn_dma_fold" (the twelve-wave feed-forward launch in its
 * paired-chunk / one-chunk form / under pp_proj_ffn_split_folded), "winograd_input_transform", "winograd_gemm_pool", "layernorm". Lets a test
 * assert WHICH kernels a launch plan ran (the reference has no counterpart: kernel selection there is cuDNN's, mmpose/models/heads/
 * hybrid_heads/probmap_head.py:261-294 and mmpretrain's VisionTransformer only name the layers). Unknown (and NULL) names count 0. Thread-safe (a mutex around the tally). */
long long pp_launch_count(const char* kernel);
int pp_reset_launch_counts(void);

/* Bytes of the caller-allocated buffers of one step of the launch plan (the library never allocates). `shape`
 * describes the step: prec = PP_PREC_*; n_img = crops x flip passes; n_tokens per image; embed / ffn widths; patch_k = 3 * patch
 * * patch; feat_h x feat_w = the backbone's output grid; heat_h x heat_w = the heatmap; deconv_channels = channels of the
 * deconvolution outputs. `buffer` = PP_WS_*; `index` selects the deconvolution (PP_WS_DECONV: 0, 1 ...) or the tower stage
 * (PP_WS_TOWER*: 0..2), 0 otherwise. Operand-format buffers are 2 bytes per element in PP_PREC_BF16 and 4 in PP_PREC_F32 /
 * PP_PREC_F16X3 (the split format is a 4-byte container). Returns < 0 (PP_ERR_INVALID_ARG) for an unknown buffer / bad shape. */
typedef struct {
    int prec, n_img, n_tokens, embed, ffn, patch_k, n_keypoints, feat_h, feat_w, heat_h, heat_w, deconv_channels;
} pp_plan_shape;
enum {
    PP_WS_PATCHES = 0,       /* (M, patch_k) operand format: im2col of the preprocessed crops, M = n_img * n_tokens        */
    PP_WS_X = 1,             /* (M, embed) fp32: the residual stream                                                       */
    PP_WS_H = 2,             /* (M, embed) operand format: LayerNorm output feeding qkv                                    */
    PP_WS_QKV = 3,           /* (M, 3 embed) operand format (unfused qkv + attention only)                                 */
    PP_WS_ATT = 4,           /* (M, embed) operand format: attention output of pp_qkv_attention_split                      */
    PP_WS_LN2 = 5,           /* (M, embed) operand format: ln2 rows parked by pp_proj_ffn_split_residual_layernorm         */
    PP_WS_FFN = 6,           /* (M, ffn) operand format: hidden activation (unfused FFN only)                              */
    PP_WS_FEAT = 7,          /* (M, embed) operand format: final LayerNorm = NHWC feature map                              */
    PP_WS_LOGITS = 8,        /* (n_img, K, heat_h * heat_w) fp32                                                           */
    PP_WS_DECONV = 9,        /* output of deconvolution `index`, NHWC operand format                                       */
    PP_WS_TOWER = 10,        /* (4, n_img, h, w, embed) operand format: convolution output of tower stage `index`         */
    PP_WS_TOWER_PARTIAL = 11,/* (slices, 4, n_img, h, w, embed) fp32: split-K partial sums of tower stage `index`, slices =
                              * pp_conv3x3_splitk_slices(prec, n_img, h, w, embed, embed, 4)                                                    */
    PP_WS_TOWER_POOLED = 12, /* (4, n_img, h / ph, w / pw, embed) operand format: pooled output of tower stage `index`     */
    PP_WS_WINOGRAD = 13,     /* pp_conv3x3_winograd_maxpool_relu's scratch for the first tower stage (pp_winograd_scratch_bytes)  */
    PP_WS_LN_STATS = 14      /* (n_img * n_tokens, embed / 96, 2) fp32: row statistics between two pp_linear_ln_folded launches   */
};
long long pp_workspace_bytes(int buffer, int index, const pp_plan_shape* shape);

/* K-slices of pp_conv3x3_splitk for a small tower stage (probmap_head.py:261-294, the 4 x 4 and 2 x 2 stages; `groups` = the four
 * towers): 3 = one kernel row of taps per slice, 9 = one tap per slice when the stage has fewer than option "ksplit9_below" output
 * rows B * H * W (too few tiles to fill the chip otherwise); 4 = four CHANNEL ranges (all nine taps each) on the split-fp16
 * wide-tile kernel, when the stage has enough rows for its 256 x 192 tiles to fill the chip (PP_PREC_F16X3, Cin % 128 == 0,
 * Cout % 192 == 0; option "ksplit_channels"). The caller passes this count to pp_conv3x3_splitk / pp_sum_maxpool_relu_nhwc;
 * PP_WS_TOWER_PARTIAL is sized for it. */
int pp_conv3x3_splitk_slices(int prec, int B, int H, int W, int Cin, int Cout, int groups);

/* Measurement aid (bench hygiene, no reference counterpart): one wavefront that sleeps on `stream` until *stop_flag becomes
 * non-zero (stop_flag: a word both sides can address - pinned host memory the caller sets with a plain store; NULL = no flag) or
 * `max_microseconds` (<= 5 000 000) of wall time have passed, then writes out[0] = shader-clock cycles (s_memtime) and out[1] =
 * ticks of the constant 100 MHz counter (s_memrealtime) it saw meanwhile: out[0] / out[1] * 100 = the average shader clock in
 * MHz over that window. Launched on a side stream beside a timed loop it records the clock the loop actually ran at (box-to-box
 * spread is clock spread). `out` may be pinned host memory too. */
int pp_clock_probe(unsigned long long* out_cycles_ticks, const unsigned long long* stop_flag, unsigned int max_microseconds, void* stream);

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

/* Same as pp_probmap_decode but starting from the LOGITS of the head's final 1x1 conv: the
 * kernel first applies  x / temperature -> Sparsemax over the H*W pixels of each (b, k) row ->
 * * normalize -> clamp(0, 1)  (probmap_head.py:637-646; Sparsemax = PyPI `sparsemax` [3P],
 * probmap_head.py:11,251), to the row itself and -- under the flip test -- to the mirror
 * partner's row, then averages and decodes as above. Logits are read from HBM once; the
 * probability maps only leave the CU when avg_out is requested. H*W <= 7168.
 * normalize < 0 stands for the head's `normalize=None` (probmap_head.py:249: normalize_layer = Identity): no Sparsemax,
 * the map is clamp(x / temperature, 0, 1). */
int pp_probmap_head_decode(const float* logits, const float* logits_flip, const int32_t* flip_indices,
                           const double* taps, const int32_t* radius,
                           int B, int K, int H, int W, double in_w, double in_h,
                           float temperature, float normalize,
                           float* avg_out, float* conv_out,
                           float* locs, double* keypoints, float* scores, void* stream);

/* ------------------------------------------------------------------------------------
 * Operand precision of the MFMA kernels.
 *   PP_PREC_BF16: bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16) -- throughput mode.
 *   PP_PREC_F32 : fp32 operands, fp32 accumulate (v_mfma_f32_16x16x4_f32, exact fp32
 *                 products) -- bit-for-bit an fp32 fmaf chain, at 1/16 of the bf16 MFMA rate.
 *   PP_PREC_F16X3: split-fp16 operands (x = hi + lo, both fp16; a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
 *                 v_mfma_f32_16x16x32_f16, fp32 accumulate): operand error ~2^-23, fp32's own rounding step --
 *                 the mode that meets the reference's 1e-3 tolerance at MFMA fp16 rate. Operand tensors are 4 bytes
 *                 per element in blocks of 32 elements along the contiguous axis: 32 hi halves (64 B) then 32 lo
 *                 halves (64 B); every row pitch / channel count must be a multiple of 32. Host-side packing:
 *                 probpose_code_amd/weights.py::to_split.
 * Output-format flags (`out_bf16`, `h_bf16`, `in_bf16`, `feat_bf16` arguments below): PP_OUT_F32 = fp32,
 * PP_OUT_BF16 = bf16 (with PP_PREC_BF16), PP_OUT_SPLIT = split fp16 (with PP_PREC_F16X3).
 * ---------------------------------------------------------------------------------- */
enum { PP_PREC_BF16 = 0, PP_PREC_F32 = 1, PP_PREC_F16X3 = 2 };
enum { PP_OUT_F32 = 0, PP_OUT_BF16 = 1, PP_OUT_SPLIT = 2 };
enum { PP_ACT_NONE = 0, PP_ACT_GELU = 1, PP_ACT_RELU = 2 };
enum { PP_CONV3X3 = 1, PP_DECONV4X4S2 = 2 };

/* Dense layer  out[m, n] = act_fn(sum_k act[m, k] * weight[n, k] + bias[n]) + residual[r(m), n].
 *
 * Stands in for every nn.Linear of the backbone (mmpretrain VisionTransformer [3P]; call site
 * mmpose/models/pose_estimators/base.py:206, ctor args config :56-67): qkv, proj, FFN.
 * act/weight are bf16 or fp32 according to `prec`; bias and residual are fp32 (or NULL);
 * `out` is bf16 when out_bf16 != 0 else fp32. K must be a multiple of 64 (bf16) / 32 (fp32);
 * lda/ldw multiples of 8 elements; ldc a multiple of 4. act_fn: PP_ACT_GELU is the exact erf form
 * (nn.GELU()). residual may alias out (in-place residual stream); r(m) = m, or m % res_mod when
 * res_mod > 0 (a (res_mod, N) table broadcast over the batch: the ViT pos_embed added to the
 * patch-embed output). planar_P > 0 stores fp32 planes out[((m / P) * N + n) * P + m % P]
 * instead of rows: the final 1x1 conv of the heatmap branch (probmap_head.py:244-249) emits
 * (B, K, H*W) logits, the layout the Sparsemax/decode kernel streams. */
int pp_gemm(int prec, const void* act, const void* weight, const float* bias, const float* residual,
            int res_mod, void* out, int M, int N, int K, int lda, int ldw, int ldc, int act_fn,
            int out_bf16, int planar_P, void* stream);
/* ... with PP_PREC_F16X3 weights stored as W * 2^e (numeric domain above): w_inv_scale = 2^-e multiplies the sums in front of the bias. A power
 * of two in [2^-40, 2^40]; 1 for the other precisions. */
int pp_gemm_ws(int prec, const void* act, const void* weight, const float* bias, const float* residual,
               int res_mod, void* out, int M, int N, int K, int lda, int ldw, int ldc, int act_fn,
               int out_bf16, int planar_P, float w_inv_scale, void* stream);

/* Dense layer of a ViT block (PP_PREC_F16X3 operands) with the LayerNorm in FRONT of it folded into the layer and the statistics of the
 * LayerNorm BEHIND it emitted with the output rows: a block of mmpretrain's TransformerEncoderLayer [3P] (x = x + attn(ln1(x));
 * x = ffn(ln2(x)) + x; call site mmpose/models/pose_estimators/base.py:206) then runs without a LayerNorm launch and with its residual
 * stream in the operand format (the path ViT-B - BASELINE config 4 - takes; ViT-S at 192 tokens has its two-launch layer kernels).
 *   out[m, n] = act_fn(sum_k a_hat[m, k] * weight[n, k] + bias[n]) + residual[m, n]
 *   a_hat[m, :] = act[m, :]                                  without ln_stats
 *               = (act[m, :] - mean[m]) * rstd[m]            with ln_stats: evaluated as rstd * (sum_k act weight - mean * ln_colsum[n]);
 *                 the CALLER folds the affine part into the layer: weight[n, k] = W[n, k] * gamma[k], ln_colsum[n] = sum_k weight[n, k]
 *                 (of the split-rounded weights), bias[n] = b[n] + sum_k W[n, k] * beta[k] (probpose_code_amd/weights.py::fold_layernorm)
 * ln_stats : (M, K / 96, 2) fp32, per row and 96-column part (mean, sum of squared deviations from it) of the act rows - what the layer
 *            that PRODUCED those rows left in its stats_out (its N = this K); mean / rstd (biased variance + ln_eps, as nn.LayerNorm)
 *            are combined from the parts here. K % 192 == 0.
 * stats_out: (M, N / 96, 2) fp32 or NULL: the same statistics of the OUTPUT rows (after activation and residual).
 * residual : NULL, fp32 rows (residual_format PP_OUT_F32) or operand-format rows (PP_OUT_SPLIT: hi + lo, 22 significant bits; may alias
 *            out - the residual stream updated in place). out_format PP_OUT_SPLIT or PP_OUT_F32. act must not alias out.
 * Shapes: K % 32 == 0, K >= 64, N % 192 == 0 (192 x 192 tiles, twelve-wave kernel of pp_linear_dma.hip); ln_stats excludes residual and
 * stats_out (the block has no such layer: qkv / fc1 take statistics, proj / fc2 make them); PP_ERR_UNSUPPORTED otherwise.
 * pp_linear_ln_folded_supported: 0 = shape not served, 1 = served, 2 = served with at least two rounds of tiles on the chip (the size from
 * which pp_gemm itself picks this kernel; callers use the folded plan from there and pp_gemm + pp_layernorm below). */
int pp_linear_ln_folded(const void* act, const void* weight, const float* bias, const void* residual, int residual_format,
                        void* out, int out_format, int M, int N, int K, int act_fn, const float* ln_stats,
                        const float* ln_colsum, float ln_eps, float* stats_out, void* stream);
/* ... with the weights stored as W' * 2^e (ln_colsum then sums the STORED weights): w_inv_scale = 2^-e. */
int pp_linear_ln_folded_ws(const void* act, const void* weight, const float* bias, const void* residual, int residual_format,
                           void* out, int out_format, int M, int N, int K, int act_fn, const float* ln_stats,
                           const float* ln_colsum, float ln_eps, float* stats_out, float w_inv_scale, void* stream);
int pp_linear_ln_folded_supported(int M, int N, int K, int with_ln_stats);

/* Dense layer of a SMALL batch (PP_PREC_F16X3), column-parallel: the launch plan of the one-image / few-person callers
 * (mmpose/apis/inference.py:161-196: one batch per image, B = number of boxes; demo/image_demo.py:36-61: one crop;
 * demo/topdown_demo_with_mmdet.py:35-41), where the layer kernels of the headline plan - 96 complete rows per workgroup, the layer's whole weight
 * set streamed through each - would run 4 workgroups on 256 CUs:
 *   out[m, n] = act_fn((sum_k act[m, k] weight[n, k]) * w_inv_scale + bias[n]) + residual[r(m), n]
 *   ln_out[m, :] = LayerNorm(out[m, :]; ln_gamma, ln_beta, ln_eps)                       when ln_out != NULL
 * act (M, K), weight (N, K) PP_OUT_SPLIT; bias fp32 or NULL; residual fp32 (M, N) - it may be `out` (fp32), the residual stream updated in place -
 * or a (res_mod, N) table (r(m) = m % res_mod: pos_embed) or NULL; out fp32 or PP_OUT_SPLIT. Tiles of 32 / 64 / 96 rows x 32 / 64 / 96 columns, the
 * shape a measured cost rule prices cheapest (rounds of one workgroup per CU x the time of a round + the LayerNorm tail of its row block;
 * pp_skinny_linear_tile returns rows * 1000 + columns). N % 32 == 0, K % 64 == 0.
 * The LayerNorm tail needs no second launch and no grid barrier: the workgroup that stores the LAST tile of a row block (a counter per block in
 * ln_counters: ceil(M / 32) int32, zero before the first launch, left at zero) normalises the block's rows - out must be fp32 then, N <= 1024, and
 * ln_out (M, N) PP_OUT_SPLIT distinct from out and act. The result does not depend on which workgroup arrives last.
 * Stands in for attn.proj / ffn.layers.0.0 / ffn.layers.1 of mmpretrain's TransformerEncoderLayer [3P] and the patch-embed projection (call site
 * mmpose/models/pose_estimators/base.py:206) at small M. */
int pp_skinny_linear(const void* act, const void* weight, const float* bias, const float* residual, int res_mod, void* out, int out_format,
                     int M, int N, int K, int act_fn, float w_inv_scale, const float* ln_gamma, const float* ln_beta, float ln_eps,
                     void* ln_out, int* ln_counters, void* stream);
int pp_skinny_linear_tile(int M, int N, int K, int with_layernorm);
/* 1x1 convolution to a few channels, planar fp32 out, of a SMALL batch on pp_skinny_linear's 32 x 32 tiles (the final layer of the heatmap branch,
 * mmpose/models/heads/hybrid_heads/probmap_head.py:244-249, 471-472): out[i, c, p] = sum_k act[i * P + p, k] weight[c, k] * w_inv_scale + bias[c],
 * c < n_valid. act (n_img * P, K) PP_OUT_SPLIT (an NHWC map); weight_padded (32 * ceil(n_valid / 32), K) PP_OUT_SPLIT with ZERO rows past n_valid,
 * bias_padded of the same padded length; out fp32 (n_img, n_valid, P). K % 64 == 0. */
int pp_skinny_conv1x1_planar(const void* act, const void* weight_padded, const float* bias_padded, float* out, int n_img, int P, int K,
                             int n_valid, float w_inv_scale, void* stream);
/* ConvTranspose2d(Cin -> Cout, k4, s2, p1, bias=False) + folded BatchNorm + ReLU of a SMALL batch (mmpose/models/heads/hybrid_heads/
 * probmap_head.py:435-472): act_nhwc (B, H, W, Cin), out_nhwc (B, 2H, 2W, Cout) PP_OUT_SPLIT; weight / bias as PP_DECONV4X4S2 of pp_conv_gemm
 * with py < 0 (four phase matrices (Cout, 4 Cin), BatchNorm folded). The four output phases are four column-parallel GEMMs of one launch on
 * pp_skinny_linear's tiles, the taps gathered by LDS-DMA (zeros outside the map); same sums in the same order as pp_conv_gemm. */
int pp_skinny_deconv(const void* act_nhwc, const void* weight, const float* bias, void* out_nhwc, int B, int H, int W, int Cin, int Cout,
                     void* stream);

/* Residual dense layer fused with the LayerNorm that follows it in the ViT block
 * (mmpretrain TransformerEncoderLayer [3P]: x = x + attn(ln1(x)); x = ffn(ln2(x)) + x; final ln1):
 *   x_out[m, :] = sum_k act[m, k] * weight[:, k] + bias + residual[r(m), :]        (fp32 residual stream)
 *   h_out[m, :] = LayerNorm(x_out[m, :]; gamma, beta, eps)                          (bf16 or fp32 operand of the next GEMM)
 * One workgroup owns 96 (N = 384, ViT-S) or 112 (N = 768, ViT-B) complete rows, so the row statistics never
 * leave the CU and the separate LayerNorm pass over the stream disappears. N must be 384 or 768; callers fall
 * back to pp_gemm + pp_layernorm otherwise. residual may alias x_out; act may alias h_out (a workgroup
 * has consumed its own rows of act before it writes them). r(m) as in pp_gemm (res_mod). */
int pp_gemm_residual_layernorm(int prec, const void* act, const void* weight, const float* bias,
                               const float* residual, int res_mod, float* x_out, const float* gamma,
                               const float* beta, float eps, void* h_out, int h_bf16, int M, int N, int K,
                               int lda, int ldw, void* stream);
/* ... with the PP_PREC_F16X3 weights stored as W * 2^e: w_inv_scale = 2^-e (residual + bias enter the accumulators times 2^e, exact). */
int pp_gemm_residual_layernorm_ws(int prec, const void* act, const void* weight, const float* bias,
                               const float* residual, int res_mod, float* x_out, const float* gamma,
                               const float* beta, float eps, void* h_out, int h_bf16, int M, int N, int K,
                               int lda, int ldw, float w_inv_scale, void* stream);

/* Whole feed-forward block of a ViT layer fused with the LayerNorm that follows it (bf16 operands):
 *   x_out = residual + GELU(h_in W1^T + b1) W2^T + b2 ;  h_out = LayerNorm(x_out; gamma, beta, eps)
 * (mmpretrain TransformerEncoderLayer.ffn [3P] = mmcv FFN: Linear(E, F) - GELU - Linear(F, E) + identity).
 * The F-wide hidden activation stays on the CU. h_in, w1 (F, E), w2 (E, F), h_out are bf16; b1, b2,
 * residual, x_out, gamma, beta fp32. E must be 384, F a multiple of 128. residual may alias x_out
 * and h_in may alias h_out. */
int pp_mlp_residual_layernorm(const void* h_in, const void* w1, const float* b1, const void* w2,
                              const float* b2, const float* residual, float* x_out, const float* gamma,
                              const float* beta, float eps, void* h_out, int M, int E, int F, void* stream);

/* Second half of a ViT layer in one launch (bf16 operands): attention output projection + residual, the LayerNorm
 * in front of the FFN, the FFN + residual, and the LayerNorm that follows the layer
 *   x1    = residual + attn Wp^T + bp ;          h  = LayerNorm(x1; gamma2, beta2, eps)
 *   x_out = x1 + GELU(h W1^T + b1) W2^T + b2 ;   h_out = LayerNorm(x_out; gamma, beta, eps)
 * (mmpretrain TransformerEncoderLayer.forward [3P]: x = x + attn(ln1(x)); x = ffn(ln2(x), identity=x), here from
 * the point where the per-head attention outputs exist; ln1 of the NEXT layer / the final norm is the trailing
 * LayerNorm). x1 and h never leave the CU. attn (M, E), wp (E, E), w1 (F, E), w2 (E, F), h_out are bf16; the
 * rest fp32. E must be 384, F a multiple of 128. residual may alias x_out.
 * With wqkv != NULL the next layer's qkv Linear (nn.Linear(E, 3E), weight (3E, E) bf16, bias fp32) is applied to
 * h_out in the same launch:  qkv_out (M, 3E) bf16 = h_out wqkv^T + bqkv ; h_out may then be NULL (not stored). */
int pp_proj_mlp_residual_layernorm(const void* attn, const void* wp, const float* bp, const float* residual,
                                   const float* gamma2, const float* beta2, const void* w1, const float* b1,
                                   const void* w2, const float* b2, float* x_out, const float* gamma,
                                   const float* beta, float eps, void* h_out, const void* wqkv, const float* bqkv,
                                   void* qkv_out, int M, int E, int F, void* stream);

/* Convolutions of ProbMapHead as implicit GEMMs on NHWC activations (no im2col buffer):
 *   PP_CONV3X3     : Conv2d(Cin->Cout, k3, s1, p1) of the scalar towers
 *                    (mmpose/models/heads/hybrid_heads/probmap_head.py:261-410);
 *                    weight[n, (ky*3+kx)*Cin + c] = w_torch[n, c, ky, kx]; out NHWC (B,H,W,Cout).
 *   PP_DECONV4X4S2 : one output phase (py, px) of ConvTranspose2d(Cin->Cout, k4, s2, p1)
 *                    (probmap_head.py:435-472): writes out[b, 2y+py, 2x+px, :] of a (B,2H,2W,Cout)
 *                    tensor; weight[n, (ty*2+tx)*Cin + c] = w_torch[c, n, 3-2ty-py, 3-2tx-px].
 *                    py < 0 runs all four phases in one launch: `weight` then holds the four phase
 *                    matrices back to back, order (py, px) = (0,0), (0,1), (1,0), (1,1).
 * BatchNorm is folded into weight/bias by the caller; act_fn applies after bias. `groups`
 * launches several independent convolutions at once (the four towers): operand g is at
 * base + g * stride_*_g elements. ldc = row stride of out in elements. */
int pp_conv_gemm(int prec, int kind, const void* act_nhwc, const void* weight, const float* bias, void* out,
                 int B, int H, int W, int Cin, int Cout, int py, int px, int groups,
                 long long stride_act_g, long long stride_w_g, long long stride_out_g,
                 long long stride_bias_g, int ldc, int act_fn, int out_bf16, void* stream);

/* Multi-head self-attention of the backbone (mmpretrain MultiheadAttention [3P]:
 * softmax(q k^T * scale) v per head). qkv: (n_seq * seq_len, 3 * heads * head_dim) rows as the
 * qkv Linear emits them ([q | k | v], head-major inside each); out: (n_seq * seq_len, heads *
 * head_dim). Both bf16 or fp32 per `prec`. Supported (seq_len, head_dim): {192, 432} x {32, 64}
 * (fp32 432 x 64 exceeds one CU's LDS). */
int pp_attention(int prec, const void* qkv, void* out, int n_seq, int seq_len, int heads, int head_dim,
                 float scale, void* stream);

/* Input side of the backbone: PoseDataPreprocessor (mmpose/models/data_preprocessors/
 * data_preprocessor.py:79-104 + mmengine ImgDataPreprocessor [3P]: BGR->RGB, float, (x-mean)/std),
 * the flip-test copy `inputs.flip(-1)` (mmpose/models/pose_estimators/topdown.py:109-112) and the
 * im2col of the ViT patch-embed Conv2d(3->E, k16, s16, zero pad `pad`) in one pass.
 * img: (B, 3, H, W) CHW, uint8 (raw crops; normalised here) or -- img_is_f32 != 0 -- fp32 that is
 * already preprocessed (the reference backbone's own input contract; mean/std/bgr_to_rgb ignored).
 * patches: (passes * B * Hp * Wp, 768) bf16/fp32, row order (pass, b, py, px), column order
 * (c, i, j); pass 1 is the horizontally flipped crop.
 * mean_host / std_host: 3 floats each, HOST pointers, in output-channel (RGB) order. */
int pp_preproc_im2col(int prec, const void* img, int img_is_f32, void* patches, int B, int passes, int H,
                      int W, int patch, int pad, const float* mean_host, const float* std_host,
                      int bgr_to_rgb, void* stream);

/* LayerNorm over the last dim of an fp32 (M, E) matrix (nn.LayerNorm(E, eps) of the ViT, eps 1e-6);
 * output bf16 or fp32. E in {384, 768, 1024}. */
int pp_layernorm(const float* x, const float* gamma, const float* beta, void* y, int M, int E, float eps,
                 int out_bf16, void* stream);

/* MaxPool2d(kernel = stride = (ph, pw)) + ReLU on an NHWC tensor (the towers' pooling,
 * probmap_head.py:264,277-278): (N, H, W, C) -> (N, H/ph, W/pw, C). */
int pp_maxpool_relu_nhwc(const void* in, int in_bf16, void* out, int out_bf16, int N, int H, int W, int C,
                         int ph, int pw, void* stream);

/* First stage of the four scalar towers in one launch: Conv2d(Cin->Cout, k3, p1) + folded BN -> MaxPool2d(ph, pw) -> ReLU
 * (probmap_head.py:261-278), i.e. PP_CONV3X3 of pp_conv_gemm followed by pp_maxpool_relu_nhwc with the full-resolution
 * tensor never stored: out_pooled (groups, B, H/ph, W/pw, Cout). One launch for bf16 operands on 16x12 feature maps with
 * (4, 3) windows (the halo-staged kernel holds whole images per tile); every other shape / precision runs the two entry points
 * through scratch_full (groups, B, H, W, Cout), which may be NULL only when the fused form applies. fmt: PP_OUT_*. */
int pp_conv3x3_maxpool_relu(int prec, const void* act_nhwc, const void* weight, const float* bias, void* out_pooled,
                            void* scratch_full, int B, int H, int W, int Cin, int Cout, int ph, int pw, int groups,
                            long long stride_act_g, long long stride_w_g, long long stride_bias_g, int fmt, void* stream);

/* A whole ViT-S encoder layer in one launch (bf16 operands), from the qkv of this layer to the qkv of the next:
 *   a     = softmax(q k^T * scale) v per head                  (mmpretrain MultiheadAttention [3P], as pp_attention)
 *   x1    = residual + a Wp^T + bp ;          h  = LayerNorm(x1; gamma2, beta2, eps)
 *   x_out = x1 + GELU(h W1^T + b1) W2^T + b2 ;   h_out = LayerNorm(x_out; gamma, beta, eps)
 *   qkv_out = h_out wqkv^T + bqkv                               (when wqkv != NULL; h_out may then be NULL)
 * i.e. pp_attention followed by pp_proj_mlp_residual_layernorm with neither the attention output nor anything between
 * the two residual adds leaving the CU. qkv_in (M, 3E) bf16 row-major [q | k | v] as the qkv Linear emits it; rows
 * [s * seq_len, (s + 1) * seq_len) are sequence s. Built for E = 384, heads * 32 = E, seq_len = 192 (256x192 input);
 * other shapes take the separate entry points. qkv_out must not alias qkv_in. */
int pp_vit_layer(const void* qkv_in, int seq_len, int heads, float scale, const void* wp, const float* bp,
                 const float* residual, const float* gamma2, const float* beta2, const void* w1, const float* b1,
                 const void* w2, const float* b2, float* x_out, const float* gamma, const float* beta, float eps,
                 void* h_out, const void* wqkv, const float* bqkv, void* qkv_out, int M, int E, int F, void* stream);

/* Whole feed-forward block of a ViT layer fused with the LayerNorm that follows it, in the parity precision
 * (PP_PREC_F16X3: split-fp16 operands, three fp16 MFMAs per product):
 *   x_out = residual + GELU(h_in W1^T + b1) W2^T + b2 ;  h_out = LayerNorm(x_out; gamma, beta, eps)
 * (mmpretrain TransformerEncoderLayer.ffn [3P] = mmcv FFN: Linear(E, F) - GELU - Linear(F, E) + identity; call site
 * mmpose/models/pose_estimators/base.py:206, ctor args td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67).
 * The F-wide hidden activation stays on the CU. h_in, h_out (M, E) are PP_OUT_SPLIT tensors; b1, b2, residual, x_out,
 * gamma, beta fp32. The weights come as ONE buffer in the kernel's consumption order, produced once per layer by
 * pp_ffn_split_pack_weights from w1 (F, E) and w2 (E, F) in the split format (pp_ffn_split_packed_bytes(E, F) bytes;
 * -1 for an unsupported shape). E must be 384, F a multiple of 128. residual may alias x_out, h_in may alias h_out
 * (a workgroup has consumed its own rows before it writes them). */
long long pp_ffn_split_packed_bytes(int E, int F);
int pp_ffn_split_pack_weights(const void* w1_split, const void* w2_split, void* packed, int E, int F, void* stream);
int pp_ffn_split_residual_layernorm(const void* h_in, const void* w_packed, const float* b1, const float* b2,
                                    const float* residual, float* x_out, const float* gamma, const float* beta,
                                    float eps, void* h_out, int M, int E, int F, void* stream);
/* ... with W1 / W2 packed from tensors stored as W * 2^e: their inverse scales (powers of two). */
int pp_ffn_split_residual_layernorm_ws(const void* h_in, const void* w_packed, const float* b1, const float* b2,
                                       const float* residual, float* x_out, const float* gamma, const float* beta,
                                       float eps, void* h_out, int M, int E, int F, float w1_inv_scale, float w2_inv_scale, void* stream);

/* The first half of a ViT layer in ONE launch in the parity precision (PP_PREC_F16X3): the qkv Linear layer and the
 * multi-head self-attention behind it; the (M, 3E) qkv tensor never reaches HBM:
 *   qkv = h_in Wqkv^T + bqkv ;  out[:, head] = softmax(q_head k_head^T * scale) v_head   per sequence
 * (mmpretrain MultiheadAttention.forward [3P]: self.qkv(x).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4), scaled dot-product
 * attention, heads concatenated; call site mmpose/models/pose_estimators/base.py:206, ctor args
 * td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67). One workgroup per (sequence, head). h_in, out (n_seq * seq_len, E)
 * PP_OUT_SPLIT; wqkv (3E, E) in the split format, rows [q | k | v] as mmpretrain packs them; bqkv (3E) fp32 or NULL.
 * Built for seq_len 192, head_dim 32, heads * head_dim = 384 (ViT-S at 256x192); PP_ERR_UNSUPPORTED otherwise (callers then
 * use pp_gemm + pp_attention). out must not alias h_in. */
int pp_qkv_attention_split(const void* h_in, const void* wqkv, const float* bqkv, void* out, int n_seq, int seq_len, int heads,
                           int head_dim, float scale, void* stream);
/* ... with Wqkv stored as W * 2^e: w_inv_scale = 2^-e. */
int pp_qkv_attention_split_ws(const void* h_in, const void* wqkv, const float* bqkv, void* out, int n_seq, int seq_len, int heads,
                              int head_dim, float scale, float w_inv_scale, void* stream);

/* The same launch with the LayerNorm in front of it (ln1 of the block; mmpretrain TransformerEncoderLayer [3P]: x + attn(ln1(x))) folded into the
 * projection: x_centered holds the residual rows MINUS THEIR ROW MEAN in the operand format, wqkv_folded / bqkv_folded carry gamma / beta
 * (probpose_code_amd/weights.py::fold_layernorm: W' = W gamma * 2^e, b' = b + W beta), and ln_stats holds (mean, rstd) of every row -
 * (n_seq * seq_len, 2) fp32, as pp_proj_ffn_split_folded leaves both. q / k / v are evaluated as  rstd * w_inv_scale * ((x - mean) W'^T) + b'
 * where the unfolded launch adds its bias: no  mean * colsum  correction, hence the accuracy of a LayerNorm applied before the split (ABI 3 took raw
 * rows and column sums: 2 - 5x the error at |mean| / std <= 1, a digit more per decade beyond). Same shapes and restrictions as
 * pp_qkv_attention_split. */
int pp_qkv_attention_split_folded(const void* x_centered, const void* wqkv_folded, const float* bqkv_folded, const float* ln_stats, void* out,
                                  int n_seq, int seq_len, int heads, int head_dim, float scale, float w_inv_scale, void* stream);

/* The second half of a ViT layer in ONE launch in the parity precision (PP_PREC_F16X3): attention output projection +
 * residual, ln2, the feed-forward block + residual, and the LayerNorm that follows the layer:
 *   x_mid = residual + att Wp^T + bp ;  h = LayerNorm(x_mid; gamma2, beta2, eps)
 *   x_out = x_mid + GELU(h W1^T + b1) W2^T + b2 ;  h_out = LayerNorm(x_out; gamma, beta, eps)
 * (mmpretrain TransformerEncoderLayer.forward [3P]: x = x + attn(ln1(x)); x = ffn(ln2(x), identity=x); call site
 * mmpose/models/pose_estimators/base.py:206, ctor args td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67).
 * x_mid and the hidden activation never leave the CU; h (the ln2 rows) passes through h_scratch (M, E; PP_OUT_SPLIT), which
 * each workgroup writes and streams back for its own 96 rows (it stays in L2) - h_scratch must not alias att or h_out.
 * att, h_out (M, E) PP_OUT_SPLIT; wproj_packed from pp_proj_split_pack_weights (Wp (E, E) in the split format;
 * pp_proj_split_packed_bytes(E) bytes, -1 if unsupported), w_packed from pp_ffn_split_pack_weights. E must be 384, F a
 * multiple of 128. residual may alias x_out, att may alias h_out. */
long long pp_proj_split_packed_bytes(int E);
int pp_proj_split_pack_weights(const void* wp_split, void* packed, int E, void* stream);
int pp_proj_ffn_split_residual_layernorm(const void* att, const void* wproj_packed, const float* bproj,
                                         const float* gamma2, const float* beta2, void* h_scratch, const void* w_packed,
                                         const float* b1, const float* b2, const float* residual, float* x_out,
                                         const float* gamma, const float* beta, float eps, void* h_out, int M, int E,
                                         int F, void* stream);
/* ... with Wp / W1 / W2 packed from tensors stored as W * 2^e: their inverse scales (powers of two). */
int pp_proj_ffn_split_residual_layernorm_ws(const void* att, const void* wproj_packed, const float* bproj,
                                            const float* gamma2, const float* beta2, void* h_scratch, const void* w_packed,
                                            const float* b1, const float* b2, const float* residual, float* x_out,
                                            const float* gamma, const float* beta, float eps, void* h_out, int M, int E,
                                            int F, float wp_inv_scale, float w1_inv_scale, float w2_inv_scale, void* stream);

/* The same launch inside a chain of layers whose ln1 is folded into the qkv projection (pp_qkv_attention_split_folded). Rows that travel between
 * the launches of the chain are CENTERED rows: x - mean(x) in the operand format, with (mean, rstd) per row beside them.
 *   residual_format PP_OUT_SPLIT: `residual` holds centered rows, residual_stats ((M, 2) fp32: mean, rstd) their means - x = (hi + lo) + mean;
 *                   PP_OUT_F32: fp32 rows (the first layer: the patch embedding's), residual_stats unused;
 *   fold_out != 0: the LayerNorm behind the FFN is NOT applied - the new rows leave ONCE, centered, in the operand format, to h_out (which may alias
 *                  `residual`), with (mean, rstd) of every row in stats_out ((M, 2) fp32; may alias residual_stats: a workgroup owns its rows);
 *                  x_out, gamma, beta are not used. A workgroup writes 288 KiB less per 96 rows (the store path is what the epilogue waits for);
 *   fold_out == 0: x_out (fp32) and h_out = LayerNorm(x_out; gamma, beta) as in the plain launch (the last layer: ln_f).
 *   w*_inv_scale:  inverse power-of-two scales of Wp / W1 / W2 (1 for unscaled weights).
 * Paired kernel only: F / 128 even - PP_ERR_UNSUPPORTED otherwise (callers keep the plain launches). Tallied as "ffn_dma_fold". */
int pp_proj_ffn_split_folded(const void* att, const void* wproj_packed, const float* bproj, const float* gamma2, const float* beta2,
                             void* h_scratch, const void* w_packed, const float* b1, const float* b2, const void* residual,
                             int residual_format, const float* residual_stats, int fold_out, float* x_out, const float* gamma,
                             const float* beta, float eps, void* h_out, float* stats_out, int M, int E, int F, float wp_inv_scale,
                             float w1_inv_scale, float w2_inv_scale, void* stream);

/* Last deconvolution of the heatmap branch fused with the 1x1 convolution that follows it (bf16 operands):
 *   ConvTranspose2d(Cin -> 256, k4, s2, p1) + BN + ReLU  ->  Conv2d(256 -> K, k1)
 * (probmap_head.py:435-472 and :244-249; reshaped to (B, K, H'W') at :627-648). weight / bias as PP_DECONV4X4S2 of
 * pp_conv_gemm with py < 0 (four phase matrices, folded BatchNorm); head_w (32, 256) bf16 = the 1x1 kernel, rows >= K
 * zero; head_b (K) fp32, K <= 28 (the kernel keeps both biases in the unused weight rows of its LDS image). The 256-channel
 * feature map is never stored. logits_phased (B, K, 4, H*W) fp32: output pixel
 * (2y + py, 2x + px) of map (b, k) sits at [b, k, 2 py + px, y W + x] - the layout pp_probmap_head_decode_phased reads. */
int pp_deconv_head(const void* act_nhwc, const void* weight, const float* bias, const void* head_w, const float* head_b,
                   float* logits_phased, int B, int H, int W, int Cin, int Cout, int K, void* stream);

/* pp_deconv_head in the parity precision (PP_PREC_F16X3): act_nhwc, weight in the split format; head_w_packed = the 1x1
 * kernel (K <= 28 maps x 256 channels, zero-padded to 32 rows) as the 32 KiB register image the kernel's waves load
 * (probpose_code_amd/weights.py::pack_head_split: [column group 4][K block 2][map fragment 2][hi | lo][lane 64][8 halves]).
 * The ReLU'd tile is split in registers and contracted with it in the epilogue; same logits_phased layout. Needs
 * 4 * ceil(B H W / 192) >= 192 tiles (PP_ERR_UNSUPPORTED below that: use pp_conv_gemm + pp_gemm). */
int pp_deconv_head_split(const void* act_nhwc, const void* weight, const float* bias, const void* head_w_packed,
                         const float* head_b, float* logits_phased, int B, int H, int W, int Cin, int Cout, int K, void* stream);

/* pp_probmap_head_decode for logits in the phase-separated layout of pp_deconv_head (H, W = the heatmap size). */
int pp_probmap_head_decode_phased(const float* logits, const float* logits_flip, const int32_t* flip_indices,
                                  const double* taps, const int32_t* radius, int B, int K, int H, int W, double in_w,
                                  double in_h, float temperature, float normalize, float* avg_out, float* conv_out,
                                  float* locs, double* keypoints, float* scores, void* stream);

/* The three decode entry points above as one, with flags: PP_DECODE_LOGITS (input = logits of the final 1x1 conv: temperature,
 * Sparsemax, normalize, clamp first, as pp_probmap_head_decode; otherwise probability maps as pp_probmap_decode and temperature /
 * normalize are ignored), PP_DECODE_PHASED (logits in the phase-separated layout of pp_deconv_head), PP_DECODE_SHIFT_HEATMAP
 * (flip_heatmaps(..., shift_heatmap=True), mmpose/models/utils/tta.py:64-66: the flipped-back map is moved one pixel to the right
 * before the average - column x takes the flipped pass's column W - x, column 0 its column W - 1; ignored without a flipped pass). */
#define PP_DECODE_LOGITS 1
#define PP_DECODE_PHASED 2
#define PP_DECODE_SHIFT_HEATMAP 4
int pp_probmap_decode_flags(const float* maps, const float* maps_flip, const int32_t* flip_indices, const double* taps,
                            const int32_t* radius, int B, int K, int H, int W, double in_w, double in_h, float temperature,
                            float normalize, float* avg_out, float* conv_out, float* locs, double* keypoints, float* scores,
                            int flags, void* stream);

/* Split-K form of the towers' 3x3 convolution for stages with few output pixels: the contraction is cut into ksplit slices -
 * 1, 3 or 9: whole taps (any precision); any other count with Cin % (32 ksplit) == 0: channel ranges [s Cin / ksplit, ...) of all nine
 * taps (PP_PREC_F16X3 on the wide-tile kernel only, needs >= 192 tiles of 256 x 192; pp_conv3x3_splitk_slices returns such a
 * count only where it applies) - every (slice, group) pair is its own set of output tiles, and the fp32 partial sums go to
 * partials (ksplit, groups, B*H*W, Cout) WITHOUT bias. Weight layout as PP_CONV3X3 of pp_conv_gemm. Followed by
 * pp_sum_maxpool_relu_nhwc, which reduces the slices, adds the (folded BatchNorm) bias, pools and applies ReLU
 * (probmap_head.py:261-294). */
int pp_conv3x3_splitk(int prec, const void* act_nhwc, const void* weight, float* partials, int B, int H, int W, int Cin,
                      int Cout, int groups, long long stride_act_g, long long stride_w_g, int ksplit, void* stream);
int pp_sum_maxpool_relu_nhwc(const float* partials, int nsplit, long long split_stride, const float* bias,
                             int images_per_group, void* out, int out_bf16, int N, int H, int W, int C, int ph, int pw,
                             void* stream);

/* First stage of the scalar towers in its Winograd F(2x2, 3x3) form (PP_PREC_F16X3 only; split-fp16 operands):
 *     Conv2d(Cin -> Cout, k3, p1) + folded BatchNorm -> MaxPool2d(pool_h, pool_w) -> ReLU      (probmap_head.py:261-294)
 * for `groups` towers that SHARE the input act_nhwc (B, H, W, Cin), in two launches: the input transform (every 4 x 4 patch ->
 * 16 planes [p][B * 48 tiles][Cin] in v_scratch, pp_winograd_scratch_bytes) and 16 position GEMMs with the output transform,
 * pooling, bias and ReLU in the epilogue - 2.25x fewer MFMAs than the implicit GEMM (pp_conv3x3_maxpool_relu), only the pooled
 * map is stored. u_packed: (groups, 16, Cout, Cin) split format, U_p = (G g G^T)_p of the BatchNorm-folded 3 x 3 weights
 * g (Cout, Cin, 3, 3), G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], p = 4 a + b (probpose_code_amd/weights.py computes it in
 * fp64). bias (groups, Cout) fp32; out_pooled (groups, B, H / pool_h, W / pool_w, Cout) split format. Built for H % 4 == 0, W % 6 == 0
 * (16 x 12, 24 x 18), pooling (4, 3), Cin % 128 == 0, Cout % 96 == 0; anything else returns PP_ERR_UNSUPPORTED (use pp_conv3x3_maxpool_relu). */
long long pp_winograd_scratch_bytes(int B, int H, int W, int Cin);
int pp_conv3x3_winograd_maxpool_relu(const void* act_nhwc, const void* u_packed, const float* bias, void* v_scratch,
                                     void* out_pooled, int B, int H, int W, int Cin, int Cout, int pool_h, int pool_w, int groups,
                                     void* stream);

/* Last layer of the four scalar towers (Conv1x1 -> Sigmoid; ReLU for the error tower,
 * probmap_head.py:280-290,405) on the 1x1 pooled feature, fused with the flip-test average of the
 * scalars (probmap_head.py:766-774). feat: (4, passes * B, C); w: (4, K, C) fp32; bias: (4, K);
 * out: (4, B, K) fp32 in tower order probability, visibility, oks, error; the error row is divided
 * by err_div (pass 1 to keep it raw). */
int pp_tower_final(const void* feat, int feat_bf16, const float* w, const float* bias,
                   const int32_t* flip_indices, float* out, int B, int passes, int C, int K, float err_div,
                   void* stream);

/* Ex-OKS similarity of one (image, category) cell (COCOeval.computeExtendedOks,
 * mmpose/evaluation/metrics/_cocoeval.py:540-707, iouType "keypoints"; window from fix_bbox_aspect_ratio,
 * mmpose/structures/keypoint/keypoints_min_padding.py:68-133). float64 device arrays: gt_kpts (G, K, 3) = x, y, v with
 * v = 3 for keypoints outside the activation window; gt_bbox (G, 4) xywh; gt_area (G); dt_kpts (D, K, 3) = x, y, presence
 * probability, ALREADY in evaluation order (stable descending score, at most maxDets = 20); sigmas (K);
 * gt_visibilities (n_vis) int32 = the visibility value of levels 1..n_vis (level 0 is v > 0).
 * confidence_thr = NaN stands for the reference's None. out (n_vis + 1, D, G) float64. */
int pp_extended_oks(const double* gt_kpts, const double* gt_bbox, const double* gt_area, const double* dt_kpts,
                    const double* sigmas, const int* gt_visibilities, int G, int D, int K, int n_vis,
                    double confidence_thr, double padding, int use_area, int original, double* out, void* stream);

/* Result record of the multi-GPU exchange: records[i] = [x, y, conf, prob, vis, oks, err] (float64 x 7) for the n =
 * crops x keypoints outputs - keypoints (n, 2) float64, scores (n) float32 from the decode, scalars (4, n) float32 from
 * pp_tower_final. One fixed-layout all_gather of this replaces mmengine's pickled collect_results. */
int pp_pack_records(const double* keypoints, const float* scores, const float* scalars, double* records, int n, void* stream);

/* Person heatmaps back on the image, merged: out[k, y, x] = max over the n persons of cv2.warpAffine(heatmap_n[k], M_n,
 * (img_w, img_h), INTER_LINEAR) with zero border (revert_heatmap + the np.max of merge_data_samples,
 * mmpose/structures/utils.py:105-123, 146-175) in one launch. heatmaps (n, K <= 32, hm_h, hm_w) float32; inverse_maps
 * (n, 2, 3) float64 = the image -> heatmap maps (warpAffine inverts M itself); out (K, img_h, img_w) float32. */
int pp_revert_heatmaps_max(const float* heatmaps, const double* inverse_maps, float* out, int n, int K, int hm_h, int hm_w,
                           int img_h, int img_w, void* stream);

/* In place heatmaps[k] = heatmaps[k] / sum(heatmaps[k]) * presence[k] - the posterior the visualiser draws
 * (mmpose/visualization/local_visualizer.py:827-837). heatmaps (K, H, W) float32, presence (K) float32, scratch: K * 64
 * float64. */
int pp_heatmap_posterior(float* heatmaps, const float* presence, double* scratch, int K, int H, int W, void* stream);

/* pp_extended_oks for every (image, category) cell of a dataset in one launch. Instances / detections of cell c are rows
 * [cell_gt_off[c], cell_gt_off[c + 1]) / [cell_dt_off[c], cell_dt_off[c + 1]) of the flat arrays (detections of a cell in
 * evaluation order, at most maxDets); its (n_vis + 1, Dc, Gc) float64 block is written at out + cell_out_off[c]. */
int pp_exoks_cells(const double* gt_kpts, const double* gt_bbox, const double* gt_area, const double* dt_kpts,
                   const double* sigmas, const int* gt_visibilities, const int* cell_gt_off, const int* cell_dt_off,
                   const long long* cell_out_off, int n_cells, int K, int n_vis, double confidence_thr, double padding,
                   int use_area, int original, double* out, void* stream);

/* Greedy detection <-> instance matching of every (cell, visibility level, area range) at every similarity threshold
 * (COCOeval.evaluateImg, mmpose/evaluation/metrics/_cocoeval.py:709-887, the return_matching=False branch: by similarity,
 * or by nearest box centre when match_by_bbox). ious / offsets as written by pp_exoks_cells; gt_ignore (N_gt, L) uint8 =
 * gt["ignore"][level]; gt_area (N_gt) = the area of the range test (:729-733); boxes xywh float64; area_rng (A, 2);
 * iou_thrs (T <= 64). max_gt_per_cell sizes the LDS. Outputs: dt_match (L, A, T, N_dt) int32 = row of the matched instance
 * or -1; dt_ignore (L, A, T, N_dt) uint8; gt_match (L, A, T, N_gt) int32 = row of the matching detection or -1;
 * gt_ignore_out (L, A, N_gt) uint8 = the cell's _ignore flag; sim_sum / sim_cnt (L, A, n_cells) = sum / number of the
 * similarities of the matches made over all thresholds (COCOeval.loc_similarities, :857). */
int pp_exoks_match(const double* ious, const int* cell_gt_off, const int* cell_dt_off, const long long* cell_iou_off,
                   const unsigned char* gt_ignore, const unsigned char* gt_iscrowd, const double* gt_area,
                   const double* gt_bbox, const double* dt_area, const double* dt_bbox, const double* area_rng,
                   const double* iou_thrs, int n_cells, int max_gt_per_cell, int N_gt, int N_dt, int L, int A, int T,
                   int match_by_bbox, int* dt_match, unsigned char* dt_ignore, int* gt_match,
                   unsigned char* gt_ignore_out, double* sim_sum, int* sim_cnt, void* stream);

/* Precision / recall / score tables of the dataset (COCOeval.accumulate, _cocoeval.py:889-1009; one category, one maxDets).
 * order (N_dt) int32 = np.argsort(-score, kind="mergesort") over all detections in cell order; rec_thrs (R <= 128);
 * chunk_scratch: int32 (L * A * T, ceil(N_dt / 256), 2). precision / scores (T, L, R, A) and recall (T, L, A) float64 must be
 * pre-filled with -1 by the caller: entries of rows without a counted instance are left untouched (:941-943, :955-956). */
int pp_exmap_accumulate(const int* dt_match, const unsigned char* dt_ignore, const unsigned char* gt_ignore,
                        const int* order, const double* dt_score, const double* rec_thrs, int n_cells, int N_gt, int N_dt,
                        int L, int A, int T, int R, int* chunk_scratch, double* precision, double* recall, double* scores,
                        void* stream);

/* Person crops of the top-down pipeline: n times cv2.warpAffine(img, M_i, (out_w, out_h), flags=INTER_LINEAR), zero border
 * (TopdownAffine.transform, mmpose/datasets/transforms/topdown_transforms.py:118-126), written as the CHW uint8 tensors
 * PackPoseInputs emits (mmpose/datasets/transforms/formatting.py:14-36,195). img_hwc: (img_h, img_w, channels) uint8 on
 * the device; inverse_maps: (n, 2, 3) float64 on the device, the dst -> src maps (warpAffine inverts M itself - so does
 * probpose_code_amd.transforms.invert_affine); crops_chw: (n, channels, out_h, out_w) uint8. OpenCV's fixed-point
 * bilinear arithmetic restated from its source; cv2 is not available in the build image: parity unpinned. */
int pp_warp_affine_u8(const void* img_hwc, int img_h, int img_w, int channels, const double* inverse_maps, void* crops_chw,
                      int n, int out_h, int out_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROBPOSE_MI355X_H_ */
