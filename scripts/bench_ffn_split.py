"""dev only: the fused f16x3 FFN kernel (pp_ffn_split.hip) against the two launches it replaces, at the bs 64 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
M, E, Fd = int(os.environ.get("M", 24576)), 384, 1536
h = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
b1, b2, g, be = torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
packed = torch.empty(L.lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
L.call("pp_ffn_split_pack_weights", w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, Fd, None)
f = torch.empty(M, Fd, device="cuda"); xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda")
def fused():
    L.call("pp_ffn_split_residual_layernorm", h.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(),
           xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
def two():
    L.call("pp_gemm", 2, h.data_ptr(), w1.data_ptr(), b1.data_ptr(), None, 0, f.data_ptr(), M, Fd, E, E, E, Fd, 1, 2, 0, None)
    L.call("pp_gemm_residual_layernorm", 2, f.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), 0, xo.data_ptr(),
           g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), 2, M, E, Fd, Fd, Fd, None)
for name, run in (("fused", fused), ("fc1 + fc2_res_ln", two), ("fused", fused)):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:18s} {us:7.1f} us  {4 * M * E * Fd / us / 1e6:6.0f} TF algorithmic ({12 * M * E * Fd / us / 1e6:6.0f} TF of MFMA issue)")
