// Internal interface of the MFMA GEMM family (pp_gemm.hip).
#pragma once
#include "pp_common.h"

namespace pp {

enum { G_LINEAR = 0, G_CONV3 = 1, G_DECONV = 2 };
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };

struct GemmParams {
    const void* A;          // activations: [M, lda] row-major, or NHWC tensor for the conv gathers
    const void* W;          // weights: [N, ldw] row-major, K contiguous
    void* C;                // output: [rows, ldc]
    const float* bias;      // [N] fp32 or NULL
    const float* residual;  // fp32, same indexing as C, or NULL
    int M, N, K;
    int lda, ldw, ldc;
    int H, Wd, Cin;         // conv gathers: input spatial size and channels
    int py, px;             // deconv output phase
    int act;                // ACT_*
    int out_bf16;           // 1: store bf16, 0: store fp32
    int gather;             // G_*
    int res_mod;            // >0: residual row = m % res_mod (broadcast over the batch, e.g. pos_embed)
    int ldres;              // row stride of residual (elements); 0 -> ldc
    const void* head_w;     // wide-tile deconvolution only: [32, N] bf16 (rows >= head_n zero) of a 1x1 convolution applied to the
    const float* head_b;    // ReLU'd tile in the epilogue instead of storing it: logits (B, head_n, 4 phases, H * W) fp32 to head_out
    float* head_out;
    int head_n;
    int ksplit;             // >1: split-K launch - `groups` counts (K slice, problem) pairs, z = slice * (groups / ksplit) + problem;
                            // K is the length of one slice, ldw the full row; conv gathers start at tap slice * K / Cin
    int groups;             // independent problems in one launch (the four towers); tiles of all groups share the persistent grid
    int pool_h, pool_w;     // pp_conv_halo.hip only, > 0: C is the MaxPool2d(pool_h, pool_w) + ReLU of the convolution, (N, H / pool_h, W / pool_w, Cout)
    int tap_inner;          // pp_panel_split.hip: 1 = the K walk of a gathered convolution runs tap-inner (channel block -> taps) instead of tap -> channel blocks
    int tile_order;         // pp_panel_split.hip: 0 row panel -> group -> column tile (activations shared per XCD), 1 weight set -> row panels
    int planar_P;           // >0: store planar, out[((m / P) * N + n) * P + m % P]  (NHWC rows -> (B, N, P) planes)
    unsigned a_bytes, w_bytes;  // extent of the activation / weight tensor of ONE group (buffer-descriptor bound)
    float w_inv = 1.0f;     // PP_PREC_F16X3 Linear layers: the weights are stored as w * 2^e (weights.py: the tensor's largest element in [2^12, 2^13),
                            // so that the low halves of its small elements are normal fp16 numbers); the accumulators are multiplied by 2^-e = w_inv
                            // in front of bias / residual / activation (exact: a power of two)
    long long strideA_z, strideW_z, strideC_z, strideBias_z;  // grouped launch (blockIdx.z), in elements
};

int gemm(const GemmParams& p, int prec, int groups, hipStream_t s);

// pp_panel_gemm.hip: wide-tile bf16 kernel for the long-K convolutions (deconv N % 256 == 0, conv 3x3 N % 192 == 0)
bool panel_gemm_supported(const GemmParams& p, int prec, int groups);
int panel_gemm(const GemmParams& p, int groups, hipStream_t s);

// pp_conv_halo.hip: 3x3 convolution on 192-pixel images with the activations staged once per channel chunk (bf16)
bool conv_halo_supported(const GemmParams& p, int prec, int groups);
int conv_halo(const GemmParams& p, int groups, hipStream_t s);

// pp_panel_split.hip: the same wide tiles for PP_PREC_F16X3 (split-fp16 operands): Linear layers (N % 192 == 0), conv 3x3,
// deconvolutions; output split fp16 or fp32 rows (+ fp32 residual); and for the long-K bf16 Linear layers (K >= 1536)
bool panel_split_supported(const GemmParams& p, int prec, int groups);
int panel_split_gemm(const GemmParams& p, int prec, int groups, hipStream_t s);

// pp_linear_dma.hip: twelve-wave 192 x 192 tiles (eight computing waves + four DMA-only waves) for the large PP_PREC_F16X3 Linear layers
bool linear_dma_supported(const GemmParams& p, int prec, int groups);
int linear_dma_gemm(const GemmParams& p, hipStream_t s);

}  // namespace pp
