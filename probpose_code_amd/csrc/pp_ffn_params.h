// Parameter block of the fused f16x3 feed-forward launch: filled by the entry points in pp_ffn_split.hip, consumed by the kernels of pp_ffn_dma.hip
// (eight computing waves + four DMA waves).
#pragma once

namespace pp {
namespace ffs {

struct Params {
    const void* h;         // [M, 384] split: LayerNorm-ed block input
    const void* wpack;     // pre-packed W1 / W2 stream (pp_ffn_split_pack_weights)
    const float* b1;       // [F]
    const float* b2;       // [384]
    const float* residual; // fp32 [M, 384] (may alias x_out)
    float* x_out;          // fp32 [M, 384]
    const float* gamma;
    const float* beta;
    void* h_out;           // [M, 384] split: LayerNorm(x_out) (may alias h)
    int M, F;
    unsigned h_bytes, w_bytes;
    float eps;
    // PROJ form (attention output projection + residual + ln2 in front of the FFN): h is then a scratch tensor this kernel
    // writes (ln2 output) before it streams it back
    const void* att;       // [M, 384] split: attention output (heads concatenated)
    const void* wproj;     // pre-packed Wp stream (pp_proj_split_pack_weights)
    const float* bp;       // [384]
    const float* gamma2;   // ln2
    const float* beta2;
    unsigned att_bytes, wproj_bytes;
    // Folded form (pp_proj_ffn_split_folded, twelve-wave paired kernel only): res_split - `residual` holds operand-format rows; fold_out - the final
    // LayerNorm is NOT applied: the new residual rows leave ONCE, in the operand format, to h_out, with (mean, rstd) per row in stats_out - the next
    // layer's pp_qkv_attention_split_folded applies them (x_out, gamma, beta unused)
    // The rows that travel in the operand format are CENTERED (x - mean of the row): `res_stats` gives the residual rows' means back
    // ([M, 2]: (mean, rstd), may alias stats_out - a workgroup reads its rows' statistics before it writes them).
    int res_split, fold_out;
    float* stats_out;      // [M, 2]
    const float* res_stats;
    // Power-of-two weight scales: Wp / W1 / W2 are stored as w * s (weights.py: the tensor's largest element in [2^12, 2^13), so that the LOW halves
    // of its small elements are normal fp16 numbers, not subnormals with an absolute 2^-25 step); accumulators that start from fp32 values are
    // multiplied by s when they are loaded and by inv = 1 / s when they are read - exact both ways.
    float s_p, inv_p, s_1, inv_1, s_2, inv_2;
};

}  // namespace ffs
}  // namespace pp
