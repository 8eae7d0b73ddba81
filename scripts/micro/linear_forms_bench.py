"""dev: the f16x3 Linear layers of ViT-B at bs 64 (M = 55 296) on the wide-tile kernel (linear_dma = 0) and the twelve-wave kernel (1)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
M = 55296
g = torch.Generator().manual_seed(0)
def run(N, K, act, split, res):
    a = to_split(torch.randn(M, K, generator=g)).cuda(); w = to_split(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda(); r = torch.randn(M, N, generator=g).cuda() if res else None
    out = torch.empty(M, N, device="cuda")
    best = {}
    for rep in range(4):
        for opt in (0, 1):
            L.set_option("linear_dma", opt)
            fn = lambda: L.call("pp_gemm", 2, a.data_ptr(), w.data_ptr(), b.data_ptr(), L.ptr(r), 0, out.data_ptr(), M, N, K, K, K, N, act, 2 if split else 0, 0, None)
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): fn()
            e1.record(); torch.cuda.synchronize()
            best[opt] = min(best.get(opt, 1e9), e0.elapsed_time(e1) / 8 * 1e3)
    gf = 2.0 * M * N * K * 1e-9
    print(f"N={N} K={K} act={act} split_out={split} residual={res}: wide-tile {best[0]:.0f} us ({gf / best[0] * 1e3:.0f} TF), twelve-wave {best[1]:.0f} us ({gf / best[1] * 1e3:.0f} TF)")
run(2304, 768, 0, True, False)
run(3072, 768, 1, True, False)
run(3072, 768, 0, True, False)
run(768, 768, 0, False, True)
run(768, 3072, 0, False, True)
L.set_option("linear_dma", 1)
