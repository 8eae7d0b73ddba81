#!/usr/bin/env python
"""dev: pp_qkv_attention_split_folded at bs 64 (128 sequences) under the timing-only ablations of QKA_DBG (1 no attention phase, 2 no GEMM-phase MFMAs,
4 no DMA traffic - zeros from the bounds check; wrong results): how much of the launch is the L2 -> LDS fill that a three-heads-per-workgroup form
would cut by 44 %? Run once per library build (scripts/micro/build/lib_<tag>.so copied over the in-tree library ON THE GPU BOX)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import to_split  # noqa: E402

n_seq, S, E, H, hd = 128, 192, 384, 12, 32
M = n_seq * S
h = to_split(torch.randn(M, E)).cuda()
w = to_split(torch.randn(3 * E, E) / E ** 0.5).cuda()
b = torch.randn(3 * E).cuda() * 0.1
st = torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().cuda()
out = torch.empty(M, E, device="cuda")


def run():
    L.call("pp_qkv_attention_split_folded", h.data_ptr(), w.data_ptr(), b.data_ptr(), st.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, 1.0, L.stream_ptr())


for _ in range(10):
    run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(12):
        run()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 120)
clk = torch.zeros(4, dtype=torch.int64, device="cuda")
print(f"{sys.argv[1] if len(sys.argv) > 1 else '':10s} us per launch: min {min(ts):6.1f}  median {sorted(ts)[3]:6.1f}")
