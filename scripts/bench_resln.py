import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
M, E = 24576, 384
for K, name in ((384, "proj"), (1536, "fc2"), (768, "patch")):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(E, K, device="cuda") / K**0.5).to(torch.bfloat16)
    b = torch.randn(E, device="cuda"); x = torch.randn(M, E, device="cuda"); g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
    h = torch.empty(M, E, device="cuda", dtype=torch.bfloat16)
    def run(): L.call("pp_gemm_residual_layernorm", 0, a.data_ptr(), w.data_ptr(), b.data_ptr(), x.data_ptr(), 0, x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, h.data_ptr(), 1, M, E, K, K, K, None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byts = M * K * 2 + 2 * M * E * 4 + M * E * 2
    print(f"{name}: {ms*1e3:.1f} us  {2*M*E*K/ms/1e9:.0f} TF  {byts/ms/1e6:.0f} GB/s ({byts/ms/1e6/80:.0f}% HBM)")
