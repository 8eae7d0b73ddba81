# dev only: K loop / epilogue / gap per tile of the twelve-wave Linear kernel from s_memtime stamps (library built with -DLDM_STAMP=1 into
# scripts/micro/build/lib_lstamp.so; on the GPU box: cp scripts/micro/build/lib_lstamp.so probpose_code_amd/libprobpose_mi355x.so; python scripts/micro/ldm_stamps.py)
import sys, os, ctypes, math, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
from probpose_code_amd.weights import fold_layernorm
L = T._lib()
M, E = 55296, 768
buf = (ctypes.c_ulonglong * 128)()
L.lib.pp_dev_ldm_stamps.restype = ctypes.c_int
def report(tag):
    torch.cuda.synchronize()
    assert L.lib.pp_dev_ldm_stamps(buf) == 0
    for wg in (0, 1):
        t = [buf[wg * 64 + i] for i in range(64)]
        rows = []
        for i in range(8):
            a, b, c = t[3 * i], t[3 * i + 1], t[3 * i + 2]
            nxt = t[3 * i + 3] if i < 7 else c
            rows.append((b - a, c - b, nxt - c))
        k = sum(r[0] for r in rows[1:7]) / 6; e = sum(r[1] for r in rows[1:7]) / 6; g = sum(r[2] for r in rows[1:7]) / 6
        print(f"{tag} workgroup {'0' if wg == 0 else '100'}: K loop {k:7.0f}  epilogue {e:6.0f}  to the next tile's first stamp {g:5.0f} cycles (tiles 1 - 6 of this workgroup); first tile K loop {rows[0][0]}")
x = T._sp(torch.randn(M, E)); st = T._row_part_stats(T._unsp(x)).float().cuda()
for name, N, act in (("qkv (MODE 1, N = 2304)", 2304, 0), ("fc1 + GELU (MODE 1, N = 3072)", 3072, 1)):
    w = torch.randn(N, E) / math.sqrt(E); b = torch.randn(N) * 0.1
    wf, cs, bf = [t.cuda() for t in fold_layernorm(w, b, torch.ones(E), torch.zeros(E))]
    out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        L.call("pp_linear_ln_folded", x.data_ptr(), wf.data_ptr(), bf.data_ptr(), None, 0, out.data_ptr(), 2, M, N, E, act, st.data_ptr(), cs.data_ptr(), 1e-6, None, None)
    report(name)
    del out
for name, K in (("proj (MODE 2, K = 768)", 768), ("fc2 (MODE 2, K = 3072)", 3072)):
    a = T._sp(torch.randn(M, K)); w = T._sp(torch.randn(E, K) / math.sqrt(K)); b = torch.randn(E, device="cuda"); xs = T._sp(torch.randn(M, E)); so = torch.empty(M, 8, 2, device="cuda")
    for _ in range(5):
        L.call("pp_linear_ln_folded", a.data_ptr(), w.data_ptr(), b.data_ptr(), xs.data_ptr(), 2, xs.data_ptr(), 2, M, E, K, 0, None, None, 1e-6, so.data_ptr(), None)
    report(name)
