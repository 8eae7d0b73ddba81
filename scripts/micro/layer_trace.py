"""dev only: phase time stamps of the fused layer kernel built with MLP_DBG=512 (scripts/micro/mlp_ablate.sh 512)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = 24576, 384, 1536
P = ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(here, "build", os.environ.get("LIB", "libmlp_dbg512.so")))
trace = torch.zeros(8192, dtype=torch.int64, device="cuda")
lib.pp_mlp_set_trace.argtypes = [P]; lib.pp_mlp_set_trace(trace.data_ptr())
fn = lib.pp_proj_mlp_residual_layernorm
fn.restype = ctypes.c_int
fn.argtypes = [P] * 13 + [ctypes.c_float] + [P] * 4 + [ctypes.c_int] * 3 + [P]
bf = lambda *s: (torch.randn(*s, device="cuda") / s[-1] ** 0.5).bfloat16()
a, wp, w1, w2, wq = torch.randn(M, E, device="cuda").bfloat16(), bf(E, E), bf(Fd, E), bf(E, Fd), bf(3 * E, E)
bp, b1, b2, bq = (torch.randn(n, device="cuda") for n in (E, Fd, E, 3 * E))
g2, be2, g, be = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
x = torch.randn(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda", dtype=torch.bfloat16); qo = torch.empty(M, 3 * E, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    assert fn(a.data_ptr(), wp.data_ptr(), bp.data_ptr(), x.data_ptr(), g2.data_ptr(), be2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
              b2.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), wq.data_ptr(), bq.data_ptr(), qo.data_ptr(), M, E, Fd, None) == 0
torch.cuda.synchronize()
t = trace.cpu().numpy()
names = ["prologue", "proj+ln2", "FFN (peeled A + 12 x [A|B])", "LN + x_out stores", "qkv tail", "final drain"]
for w in (0, 1):
    ph = t[w * 4096 + 200: w * 4096 + 207]
    print(f"wave {4*w}: total {ph[6]-ph[0]} ticks")
    for i, n in enumerate(names):
        print(f"   {n:30s} {ph[i+1]-ph[i]:7d}")
    print(f"      (of 'LN + x_out stores': row statistics {t[w * 4096 + 210] - ph[3]}, LayerNorm + store issue {ph[4] - t[w * 4096 + 210]})")
if os.environ.get("VIT"):  # the same through pp_vit_layer (attention phase in front): stamps of its twelve heads
    fv = lib.pp_vit_layer
    fv.restype = ctypes.c_int
    fv.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float] + [P] * 12 + [ctypes.c_float] + [P] * 4 + [ctypes.c_int] * 3 + [P]
    qi = torch.randn(M, 3 * E, device="cuda").bfloat16()
    trace.zero_()
    for _ in range(3):
        assert fv(qi.data_ptr(), 192, 12, 32 ** -0.5, wp.data_ptr(), bp.data_ptr(), x.data_ptr(), g2.data_ptr(), be2.data_ptr(), w1.data_ptr(),
                  b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), wq.data_ptr(),
                  bq.data_ptr(), qo.data_ptr(), M, E, Fd, None) == 0
    torch.cuda.synchronize()
    t = trace.cpu().numpy()
    for w in (0, 1):
        ph = t[w * 4096 + 200: w * 4096 + 207]
        hd = t[w * 4096 + 300: w * 4096 + 313]
        aw = t[w * 4096 + 320: w * 4096 + 332]
        nr = 9 if hd[10] == 0 else 12  # rounds of the round-robin attention phase / heads of the round-2 form (stale stamps cleared below)
        print(f"   wait + barrier per round {[int(aw[i]-hd[i]) for i in range(nr)]}")
        print(f"vit_layer wave {4*w}: total {ph[6]-ph[0]} ticks; prologue incl. attention {ph[1]-ph[0]}; start -> round 0 {hd[0]-ph[0]}; rounds {[int(hd[i+1]-hd[i]) for i in range(nr)]}")
        print("   phases: " + "  ".join(f"{n} {int(ph[i+1]-ph[i])}" for i, n in enumerate(names)) + f"  (row statistics {int(t[w * 4096 + 210] - ph[3])})")
    sys.exit(0)
for w in (0, 1):
    for pair in range(4):
        q = t[w * 4096 + (22 + 2 * pair) * 10: w * 4096 + (22 + 2 * pair) * 10 + 13]
        print(f"wave {4*w} blocks {2*pair},{2*pair+1}: step durations {[int(q[i+1]-q[i]) for i in range(11) if q[i+1] > 0 and q[i] > 0]}")
