import sys, os, math, torch
R = "/root/repo" if not os.environ.get("GRAFT_REPO_ROOT") else os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
from probpose_code_amd.weights import fold_layernorm
L = T._lib()
M, E = 55296, 768
x = T._sp(torch.randn(M, E)); st = T._row_part_stats(T._unsp(x)).float().cuda()
cases = []
for name, N, act in (("qkv", 2304, 0), ("fc1", 3072, 1)):
    w = torch.randn(N, E) / math.sqrt(E); b = torch.randn(N) * 0.1
    wf, cs, bf = [t.cuda() for t in fold_layernorm(w, b, torch.ones(E), torch.zeros(E))]
    out = torch.empty(M, N, device="cuda")
    cases.append((name, lambda wf=wf, bf=bf, cs=cs, out=out, N=N, act=act: L.call("pp_linear_ln_folded", x.data_ptr(), wf.data_ptr(), bf.data_ptr(), None, 0, out.data_ptr(), 2, M, N, E, act, st.data_ptr(), cs.data_ptr(), 1e-6, None, None)))
for name, K in (("proj", 768), ("fc2", 3072)):
    a = T._sp(torch.randn(M, K)); w = T._sp(torch.randn(E, K) / math.sqrt(K)); b = torch.randn(E, device="cuda"); xs = T._sp(torch.randn(M, E)); so = torch.empty(M, 8, 2, device="cuda")
    cases.append((name, lambda a=a, w=w, b=b, xs=xs, so=so, K=K: L.call("pp_linear_ln_folded", a.data_ptr(), w.data_ptr(), b.data_ptr(), xs.data_ptr(), 2, xs.data_ptr(), 2, M, E, K, 0, None, None, 1e-6, so.data_ptr(), None)))
for rep in range(2):
    for ns in (0, 300, 600, 1000, 1500):
        L.set_option("linear_stagger_ns", ns)
        line = f"stagger {ns:5d} ns x (w % 32):"
        for name, fn in cases:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            line += f"  {name} {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us"
        print(line, flush=True)
