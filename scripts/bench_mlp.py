import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
import os
M, E, Fd = 24576, 384, int(os.environ.get("FD", "1536"))
h = torch.randn(M, E, device="cuda").bfloat16(); w1 = (torch.randn(Fd, E, device="cuda") / E**0.5).bfloat16(); w2 = (torch.randn(E, Fd, device="cuda") / Fd**0.5).bfloat16()
b1 = torch.randn(Fd, device="cuda"); b2 = torch.randn(E, device="cuda"); x = torch.randn(M, E, device="cuda"); g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
ho = torch.empty_like(h)
def run(): L.call("pp_mlp_residual_layernorm", h.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"mlp fused: {ms*1e3:.1f} us  {4*M*E*Fd/ms/1e9:.0f} TF")
