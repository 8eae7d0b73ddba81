import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
M, E, Fd = 24576, 384, 1536
P = ctypes.c_void_p
for name in sys.argv[1:]:
    lib = ctypes.CDLL(name)
    fv = lib.pp_vit_layer; fv.restype = ctypes.c_int
    fv.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float] + [P] * 12 + [ctypes.c_float] + [P] * 4 + [ctypes.c_int] * 3 + [P]
    bf = lambda *s: (torch.randn(*s, device="cuda") / s[-1] ** 0.5).bfloat16()
    wp, w1, w2, wq = bf(E, E), bf(Fd, E), bf(E, Fd), bf(3 * E, E)
    bp, b1, b2, bq = (torch.randn(n, device="cuda") for n in (E, Fd, E, 3 * E))
    g2, be2, g, be = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
    x = torch.randn(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda", dtype=torch.bfloat16); qo = torch.empty(M, 3 * E, device="cuda", dtype=torch.bfloat16)
    qi = torch.randn(M, 3 * E, device="cuda").bfloat16()
    run = lambda: fv(qi.data_ptr(), 192, 12, 32 ** -0.5, wp.data_ptr(), bp.data_ptr(), x.data_ptr(), g2.data_ptr(), be2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), wq.data_ptr(), bq.data_ptr(), qo.data_ptr(), M, E, Fd, None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1)/20*1e3:.1f} us")
