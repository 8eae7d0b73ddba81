"""dev only: projection + ln2 + FFN + LayerNorm in one launch (pp_proj_ffn_split_residual_layernorm) against
pp_gemm_residual_layernorm + pp_ffn_split_residual_layernorm, at the bs 64 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
M, E, Fd = int(os.environ.get("M", 24576)), 384, 1536
att = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
wp = to_split(torch.randn(E, E) / E ** 0.5).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
bp, b1, b2, g, be = torch.randn(E).cuda() * 0.1, torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
packed = torch.empty(L.lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
L.call("pp_ffn_split_pack_weights", w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, Fd, None)
wpp = torch.empty(L.lib.pp_proj_split_packed_bytes(E) // 4, device="cuda")
L.call("pp_proj_split_pack_weights", wp.data_ptr(), wpp.data_ptr(), E, None)
xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda"); hs = torch.empty(M, E, device="cuda")
def fused():
    L.call("pp_proj_ffn_split_residual_layernorm", att.data_ptr(), wpp.data_ptr(), bp.data_ptr(), g.data_ptr(), be.data_ptr(),
           hs.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(), xo.data_ptr(), g.data_ptr(), be.data_ptr(),
           1e-6, ho.data_ptr(), M, E, Fd, None)
def two():
    L.call("pp_gemm_residual_layernorm", 2, att.data_ptr(), wp.data_ptr(), bp.data_ptr(), x.data_ptr(), 0, xo.data_ptr(),
           g.data_ptr(), be.data_ptr(), 1e-6, hs.data_ptr(), 2, M, E, E, E, E, None)
    L.call("pp_ffn_split_residual_layernorm", hs.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), xo.data_ptr(),
           xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
def ffn():
    L.call("pp_ffn_split_residual_layernorm", hs.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(),
           xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
for name, run in (("fused", fused), ("proj_ln + ffn", two), ("ffn alone", ffn), ("fused", fused)):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
