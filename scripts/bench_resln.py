"""dev only: pp_gemm_residual_layernorm (row-owner GEMM + residual + LayerNorm) over K, per precision and width; the slope
between two K values is the cost of one K-step without the prologue / epilogue. Usage: bench_resln.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
for prec, pname in ((0, "bf16"), (2, "f16x3")):
    for E in (384, 768):
        Mx = M if E == 384 else 27648
        res = {}
        for K in (384, 768, 1536, 3072):
            a32 = torch.randn(Mx, K, device="cuda"); w32 = torch.randn(E, K, device="cuda") / K**0.5
            a = a32.to(torch.bfloat16) if prec == 0 else to_split(a32)
            w = w32.to(torch.bfloat16) if prec == 0 else to_split(w32)
            b = torch.randn(E, device="cuda"); x = torch.randn(Mx, E, device="cuda"); g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
            h = torch.empty(Mx, E, device="cuda", dtype=torch.bfloat16 if prec == 0 else torch.float32)
            def run(): L.call("pp_gemm_residual_layernorm", prec, a.data_ptr(), w.data_ptr(), b.data_ptr(), x.data_ptr(), 0, x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, h.data_ptr(), 1 if prec == 0 else 2, Mx, E, K, K, K, None)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res[K] = e0.elapsed_time(e1) / 20 * 1e3
        bk = 64 if prec == 0 else 32
        nh = E // 384
        steps = lambda K: nh * K // bk
        slope = (res[3072] - res[768]) / (steps(3072) - steps(768))
        print(f"{pname} N={E} M={Mx}: " + "  ".join(f"K={K}: {res[K]:.1f} us ({2*Mx*E*K/res[K]/1e6 * (3 if prec == 2 else 1):.0f} TF MFMA-rate)" for K in res)
              + f"  | {slope*1e3:.0f} ns per K-step, fixed part {res[768] - steps(768) * slope:.1f} us")
