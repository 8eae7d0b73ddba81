"""Dev script: time pp_gemm on the path's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
prec = 0 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else 1
dt = torch.bfloat16 if prec == 0 else torch.float32
M = 24576
shapes = [(M, 1152, 384, "qkv"), (M, 384, 384, "proj"), (M, 1536, 384, "fc1"), (M, 384, 1536, "fc2"), (M, 384, 768, "patch"),
          (8192, 8192, 8192, "big") if prec == 0 else (4096, 4096, 4096, "big")]
for (m, n, k, name) in shapes:
    a = torch.randn(m, k, device="cuda").to(dt); w = (torch.randn(n, k, device="cuda") / k**0.5).to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt); bias = torch.randn(n, device="cuda")
    def run():
        L.call("pp_gemm", prec, a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, 0, out.data_ptr(), m, n, k, k, k, n, 0, int(prec == 0), 0, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 30
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:6s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:8.1f} TFLOP/s")
