"""dev only: per-step time stamps from the MLP kernel built with MLP_DBG=512 (scripts/micro/mlp_ablate.sh 512)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
d = sys.argv[1] if len(sys.argv) > 1 else "512"
M, E, Fd = 24576, 384, 1536
h = torch.randn(M, E, device="cuda").bfloat16(); w1 = (torch.randn(Fd, E, device="cuda") / E**0.5).bfloat16(); w2 = (torch.randn(E, Fd, device="cuda") / Fd**0.5).bfloat16()
b1 = torch.randn(Fd, device="cuda"); b2 = torch.randn(E, device="cuda"); x = torch.randn(M, E, device="cuda"); g = torch.ones(E, device="cuda"); be = torch.zeros(E, device="cuda")
ho = torch.empty_like(h)
trace = torch.zeros(8192, dtype=torch.int64, device="cuda")
P = ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(here, "build", f"libmlp_dbg{d}.so"))
lib.pp_mlp_set_trace.argtypes = [P]; lib.pp_mlp_set_trace(trace.data_ptr())
fn = lib.pp_mlp_residual_layernorm
fn.restype = ctypes.c_int
fn.argtypes = [P] * 9 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
for _ in range(3):
    assert fn(h.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None) == 0
torch.cuda.synchronize()
t = trace.cpu().numpy()
# layout: trace[w * 4096 + it * 10 + st]: it = 0 peeled phase A (6 steps), it >= 1: iteration it-1 (6 A + 4 B steps)
for w in (0, 1):
    tw = t[w * 4096: w * 4096 + 130].reshape(13, 10)
    print(f"wave {4*w}: loop {tw[12, 9] - tw[0, 0]} ticks from first to last step start")
    for it in (0, 1, 6, 12):
        n = 6 if it == 0 else 10
        d = [int((tw[it, st + 1] if st + 1 < n else tw[it + 1, 0]) - tw[it, st]) for st in range(n) if not (it == 12 and st == 9)]
        print(f"  it {it:2d}: step durations {d}  sum {sum(d)}")
