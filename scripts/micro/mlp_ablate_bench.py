"""dev only: time the libraries built by mlp_ablate.sh (python mlp_ablate_bench.py 0 1 2 ...)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = 24576, 384, int(os.environ.get("FD", "1536"))
h = torch.randn(M, E, device="cuda").bfloat16(); w1 = (torch.randn(Fd, E, device="cuda") / E**0.5).bfloat16(); w2 = (torch.randn(E, Fd, device="cuda") / Fd**0.5).bfloat16()
b1 = torch.randn(Fd, device="cuda"); b2 = torch.randn(E, device="cuda"); x = torch.randn(M, E, device="cuda"); g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
ho = torch.empty_like(h)
P = ctypes.c_void_p
for d in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(here, "build", f"libmlp_dbg{d}.so"))
    fn = lib.pp_mlp_residual_layernorm
    fn.restype = ctypes.c_int
    fn.argtypes = [P] * 9 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    def run():
        st = fn(h.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
        assert st == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"dbg {d:>4}: {ms*1e3:7.1f} us  ({4*M*E*Fd/ms/1e9:.0f} TF nominal)")
