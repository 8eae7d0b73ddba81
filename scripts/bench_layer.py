"""dev only: time the fused second-half-of-layer kernel (proj + ln2 + FFN + LN [+ next qkv]). LIB=<alternative .so>."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M, E, Fd = 24576, 384, 1536
P = ctypes.c_void_p
lib = ctypes.CDLL(os.environ.get("LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "probpose_code_amd", "libprobpose_mi355x.so")))
fn = lib.pp_proj_mlp_residual_layernorm
fn.restype = ctypes.c_int
fn.argtypes = [P] * 13 + [ctypes.c_float] + [P] * 4 + [ctypes.c_int] * 3 + [P]
bf = lambda *s: (torch.randn(*s, device="cuda") / s[-1] ** 0.5).bfloat16()
a, wp, w1, w2, wq = torch.randn(M, E, device="cuda").bfloat16(), bf(E, E), bf(Fd, E), bf(E, Fd), bf(3 * E, E)
bp, b1, b2, bq = (torch.randn(n, device="cuda") for n in (E, Fd, E, 3 * E))
g2, be2, g, be = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
x = torch.randn(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda", dtype=torch.bfloat16); qo = torch.empty(M, 3 * E, device="cuda", dtype=torch.bfloat16)
for qkv in (False, True):
    def run():
        st = fn(a.data_ptr(), wp.data_ptr(), bp.data_ptr(), x.data_ptr(), g2.data_ptr(), be2.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                w2.data_ptr(), b2.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(),
                wq.data_ptr() if qkv else None, bq.data_ptr() if qkv else None, qo.data_ptr() if qkv else None, M, E, Fd, None)
        assert st == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"qkv={qkv}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
