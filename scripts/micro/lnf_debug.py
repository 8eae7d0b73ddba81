import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
from probpose_code_amd.weights import fold_layernorm, to_split
L = T._lib()
M, E, N = 192, 768, 192
x = T._rand(M, E, seed=1) + 2.0
w = T._rand(N, E, seed=2, scale=1 / math.sqrt(E)); b = T._rand(N, seed=3)
g, be = torch.ones(E), torch.zeros(E)
wf, c, bf = fold_layernorm(w, b, g, be)
xs_, wf_, bf_, c_ = T._sp(x), wf.cuda(), bf.cuda(), c.cuda()
st = T._row_part_stats(x.double()).float().cuda()
out = torch.full((M, N), float("nan"), device="cuda")
L.call("pp_linear_ln_folded", xs_.data_ptr(), wf_.data_ptr(), bf_.data_ptr(), None, 0, out.data_ptr(), 0, M, N, E, 0, st.data_ptr(), c_.data_ptr(), 1e-6, None, None)
ref = torch.nn.functional.layer_norm(x.double(), (E,), None, None, 1e-6) @ w.double().t() + b.double()
o = out.cpu().double()
print("nan count", torch.isnan(o).sum().item(), "max err", (o - ref).abs().nan_to_num(0).max().item())
print(o[:4, :6]); print(ref[:4, :6])
out2 = torch.full((M, N), float("nan"), device="cuda")
L.call("pp_linear_ln_folded", xs_.data_ptr(), wf_.data_ptr(), bf_.data_ptr(), None, 0, out2.data_ptr(), 0, M, N, E, 0, None, None, 1e-6, None, None)
print("no-ln nan", torch.isnan(out2).sum().item(), (out2.cpu().double() - (x.double() @ w.double().t() + b.double())).abs().max().item())
from probpose_code_amd.weights import from_split
acc = x.double() @ from_split(wf).double().t()
for r in (0, 1, 17, 100, 191):
    A = torch.stack([acc[r], -c.double()], dim=1)
    sol = torch.linalg.lstsq(A, (o[r] - bf.double())[:, None]).solution.flatten()
    xr = x[r].double()
    print(r, "fitted rs", sol[0].item(), "rs*mu", sol[1].item(), "-> mu", (sol[1] / sol[0]).item(), " true rs", (1 / (xr.var(unbiased=False) + 1e-6).sqrt()).item(), "mu", xr.mean().item(),
          "resid", (A @ sol - (o[r] - bf.double())).abs().max().item())
