"""dev: ViT-S Linear shapes (K = 384, M = 24 576: what pp_gemm sees at bs 64 when the fused layer kernels are switched off) on the twelve-wave
kernel (pp_linear_dma.hip, option linear_dma = 1) against the overlapped-epilogue kernel (pp_linear_ovl.hip, linear_dma = 0) - ADVICE r4: the
twelve-wave kernel was measured on ViT-B shapes only.   python scripts/micro/linear_k384_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split

g = torch.Generator().manual_seed(1)
for M, N, K, act in ((24576, 1152, 384, 0), (24576, 1536, 384, 1), (55296, 1152, 384, 0), (55296, 1536, 384, 1)):
    a, w = to_split(torch.randn(M, K, generator=g)).cuda(), to_split(torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b, out = torch.randn(N, generator=g).cuda(), torch.empty(M, N, device="cuda")
    best = {}
    for rep in range(5):
        for opt in (1, 0):
            L.set_option("linear_dma", opt)
            run = lambda: L.call("pp_gemm", 2, a.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act, 2, 0, None)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            best[opt] = min(best.get(opt, 1e9), e0.elapsed_time(e1) / 20 * 1e3)
    L.set_option("linear_dma", 1)
    print(f"M {M} N {N} K {K} act {act}: twelve-wave {best[1]:.1f} us, overlapped-epilogue {best[0]:.1f} us")
