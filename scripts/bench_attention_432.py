"""dev only: split-fp16 attention at 432 tokens (ViT-B 384x288, bs 32 with flip: 64 sequences x 12 heads x 64 dims): the LDS-DMA kernel
(pp_attention_dma.hip) against the register-staged one (pp_set_option attn_dma 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
n_seq, S, H, hd = 64, 432, 12, int(os.environ.get("HD", 64))
E = H * hd
qkv = to_split(torch.randn(n_seq * S, 3 * E)).cuda(); out = torch.empty(n_seq * S, E, device="cuda")
def run(): L.call("pp_attention", 2, qkv.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
res, outs = {}, {}
for rep in range(4):
    for dma in (1, 0):
        L.set_option("attn_dma", dma)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(dma, []).append(e0.elapsed_time(e1) / 10 * 1e3)
        outs[dma] = out.clone()
L.set_option("attn_dma", 1)
from probpose_code_amd.weights import from_split
d = (from_split(outs[1].cpu()) - from_split(outs[0].cpu())).abs().max()
for dma in (1, 0): print(f"attn_dma={dma}: min {min(res[dma]):7.1f} us")
print("max |dma - staged| =", float(d))
