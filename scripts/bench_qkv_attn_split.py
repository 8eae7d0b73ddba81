"""dev only: qkv Linear + attention in one launch (pp_qkv_attention_split) against pp_gemm (qkv) + pp_attention, bs 64 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
n_seq, S, E, H, hd = int(os.environ.get("NSEQ", 128)), 192, 384, 12, 32
M = n_seq * S
h = to_split(torch.randn(M, E)).cuda(); w = to_split(torch.randn(3 * E, E) / E ** 0.5).cuda(); b = torch.randn(3 * E).cuda() * 0.1
qkv = torch.empty(M, 3 * E, device="cuda"); out = torch.empty(M, E, device="cuda")
def fused():
    L.call("pp_qkv_attention_split", h.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
def two():
    L.call("pp_gemm", 2, h.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0, qkv.data_ptr(), M, 3 * E, E, E, E, 3 * E, 0, 2, 0, None)
    L.call("pp_attention", 2, qkv.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
def fused2():
    L.set_option("qkv_attn_pair", 1); fused(); L.set_option("qkv_attn_pair", 0)
times = {"fused (one head)": [], "fused (head pair)": [], "qkv + attention": []}
for rep in range(5):
    for name, run in (("fused (one head)", fused), ("fused (head pair)", fused2), ("qkv + attention", two)):
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, ts in times.items():
    print(f"{name:18s} min {min(ts):7.1f} median {sorted(ts)[len(ts) // 2]:7.1f} us")
