"""dev only: per-step time stamps of the fused f16x3 FFN kernel (library built with -DFFS_DBG=512 by ffs_variants.sh):
python ffs_trace.py tag  -> cycles from barrier to barrier for block 0 (wave 0), split into [first half | wait at the barrier]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = 24576, 384, 1536
torch.manual_seed(0)
h = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
b1, b2, g, be = torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
P = ctypes.c_void_p
for tag in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(here, "build", f"libffs_{tag}.so"))
    lib.pp_ffn_split_packed_bytes.restype = ctypes.c_longlong
    packed = torch.empty(lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
    pk = lib.pp_ffn_split_pack_weights; pk.restype = ctypes.c_int; pk.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P]
    assert pk(w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, Fd, None) == 0
    fn = lib.pp_ffn_split_residual_layernorm; fn.restype = ctypes.c_int
    fn.argtypes = [P] * 8 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    trace = torch.zeros(4096, dtype=torch.int64, device="cuda")
    lib.pp_ffs_set_trace.argtypes = [P]; lib.pp_ffs_set_trace(trace.data_ptr())
    xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda")
    for _ in range(5):
        assert fn(h.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(), xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None) == 0
    torch.cuda.synchronize()
    for w in (0, 1):
        t = trace[w * 2048:(w + 1) * 2048].cpu().numpy()
        n = int((t != 0).sum()) // 2 * 2
        t = t[:n]
        pre, post = t[0::2], t[1::2]          # before / after the wait + barrier of each step
        steps = len(pre)
        print(f"{tag} wave {4 * w}: {steps} steps, total {int(post[-1] - pre[0])} ticks")
        wait = post - pre                      # time in wait + barrier
        work = pre[1:] - post[:-1]             # second half of step i + first half of step i + 1
        # peeled phase: 12 steps, then 20 per iteration
        def fmt(a): return " ".join(f"{int(v):5d}" for v in a)
        print("  peeled  work:", fmt(work[:12])); print("          wait:", fmt(wait[:12]))
        for it in (0, 5, 11):
            s0 = 12 + 20 * it
            print(f"  it {it:2d}   work:", fmt(work[s0:s0 + 20])); print("          wait:", fmt(wait[s0:s0 + 20]))
        wk, wt = work[12:232].reshape(11, 20), wait[12:232].reshape(11, 20)
        print(f"  loop mean per step (it 0-10): work {wk.mean():.0f} wait {wt.mean():.0f}; A-steps work {wk[:, :12].mean():.0f} wait {wt[:, :12].mean():.0f}; B-steps work {wk[:, 12:].mean():.0f} wait {wt[:, 12:].mean():.0f}")
