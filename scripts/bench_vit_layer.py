"""dev only: A/B of pp_vit_layer builds (one launch per ViT layer) - all libraries loaded once, timed in alternation over
several rounds (the first launches after a load run on ramping clocks), median per library. Usage: bench_vit_layer.py a.so b.so ..."""
import ctypes, os, statistics, sys
sys.path.insert(0, "/root/repo")
import torch
M, E, Fd = 24576, 384, 1536
P = ctypes.c_void_p
bf = lambda *s: (torch.randn(*s, device="cuda") / s[-1] ** 0.5).bfloat16()
wp, w1, w2, wq = bf(E, E), bf(Fd, E), bf(E, Fd), bf(3 * E, E)
bp, b1, b2, bq = (torch.randn(n, device="cuda") for n in (E, Fd, E, 3 * E))
g2, be2, g, be = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
x = torch.randn(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda", dtype=torch.bfloat16); qo = torch.empty(M, 3 * E, device="cuda", dtype=torch.bfloat16)
qi = torch.randn(M, 3 * E, device="cuda").bfloat16()
runs = {}
for name in dict.fromkeys(sys.argv[1:]):
    lib = ctypes.CDLL(name)
    fv = lib.pp_vit_layer; fv.restype = ctypes.c_int
    fv.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float] + [P] * 12 + [ctypes.c_float] + [P] * 4 + [ctypes.c_int] * 3 + [P]
    runs[name] = (lambda fv=fv: fv(qi.data_ptr(), 192, 12, 32 ** -0.5, wp.data_ptr(), bp.data_ptr(), x.data_ptr(), g2.data_ptr(), be2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), wq.data_ptr(), bq.data_ptr(), qo.data_ptr(), M, E, Fd, None))
times = {n: [] for n in runs}
for rnd in range(7):
    for name, run in runs.items():
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24): run()
        e1.record(); torch.cuda.synchronize()
        if rnd: times[name].append(e0.elapsed_time(e1) / 24 * 1e3)
for name, t in times.items():
    print(f"{name}: median {statistics.median(t):.1f} us  (min {min(t):.1f}, max {max(t):.1f})")
