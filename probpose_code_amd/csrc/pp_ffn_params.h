// Parameter block shared by the two forms of the fused f16x3 feed-forward launch: pp_ffn_split.hip (eight waves, role-alternating)
// and pp_ffn_dma.hip (eight computing waves + four DMA waves). Same packed weight streams, same outputs.
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
    unsigned long long* trace;  // dev only (FFS_DBG & 512): s_memtime at every barrier of block 0, waves 0 and 4
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
    int res_split, fold_out;
    float* stats_out;      // [M, 2]
};

}  // namespace ffs
}  // namespace pp
