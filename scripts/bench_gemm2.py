import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
prec, dt = 0, torch.bfloat16
shapes = [(24576, n, k) for n in (128, 256, 512, 1536) for k in (64, 128, 384, 1536)]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device="cuda").to(dt); w = (torch.randn(n, k, device="cuda") / k**0.5).to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    def run():
        L.call("pp_gemm", prec, a.data_ptr(), w.data_ptr(), None, None, 0, out.data_ptr(), m, n, k, k, k, n, 0, 1, 0, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 30
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    nb = (m // 128) * (n // 128)
    print(f"M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:8.1f} TF  blocks={nb} rounds={nb/512:.2f}")
