import sys, math, torch
sys.path.insert(0, "/root/repo")
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
M, N, K = 192 * 128 + 77, 768, 128
g = torch.Generator().manual_seed(0)
a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
r = torch.randn(M, N, generator=g)
ad, wd, rd = to_split(a).cuda(), to_split(w).cuda(), r.cuda()
for res in (False, True):
    out = torch.full((M, N), float("nan"), device="cuda")
    L.call("pp_gemm", 2, ad.data_ptr(), wd.data_ptr(), None, rd.data_ptr() if res else None, 0, out.data_ptr(), M, N, K, K, K, N, 0, 0, 0, None)
    ref = a.double() @ w.double().t() + (r.double() if res else 0)
    e = (out.cpu().double() - ref).abs()
    bad = e > 1e-4
    print("residual", res, "max err", e.max().item(), "bad", bad.sum().item(), "nan", torch.isnan(out).sum().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  rows", rows[:20].tolist(), "n rows", rows.numel(), " cols", cols[:20].tolist(), "n cols", cols.numel())
        rr, cc = bad.nonzero()[0].tolist()
        print("  first bad", rr, cc, out[rr, cc].item(), ref[rr, cc].item(), "gemm only", (a.double() @ w.double().t())[rr, cc].item(), "res", r[rr, cc].item())
