"""dev only: time stamps of the projection phase of pp_proj_ffn_split_residual_layernorm (library built with -DFFS_DBG=512 by
ffs_variants.sh): python ffs_trace_proj.py tag -> cycles of block 0, waves 0 and 4: start, 24 projection steps [wait | work],
ln2, drain, then the first FFN stamps."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = 24576, 384, 1536
torch.manual_seed(0)
att = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
wp = to_split(torch.randn(E, E) / E ** 0.5).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
bp, b1, b2, g, be = torch.randn(E).cuda() * 0.1, torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
P = ctypes.c_void_p
for tag in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(here, "build", f"libffs_{tag}.so"))
    lib.pp_ffn_split_packed_bytes.restype = ctypes.c_longlong
    packed = torch.empty(lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
    pk = lib.pp_ffn_split_pack_weights; pk.restype = ctypes.c_int; pk.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P]
    assert pk(w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, Fd, None) == 0
    wpp = torch.empty(E * E, device="cuda")
    pp_ = lib.pp_proj_split_pack_weights; pp_.restype = ctypes.c_int; pp_.argtypes = [P, P, ctypes.c_int, P]
    assert pp_(wp.data_ptr(), wpp.data_ptr(), E, None) == 0
    fn = lib.pp_proj_ffn_split_residual_layernorm; fn.restype = ctypes.c_int
    fn.argtypes = [P] * 13 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    trace = torch.zeros(4096, dtype=torch.int64, device="cuda")
    lib.pp_ffs_set_trace.argtypes = [P]; lib.pp_ffs_set_trace(trace.data_ptr())
    xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda"); hs = torch.empty(M, E, device="cuda")
    for _ in range(5):
        assert fn(att.data_ptr(), wpp.data_ptr(), bp.data_ptr(), g.data_ptr(), be.data_ptr(), hs.data_ptr(), packed.data_ptr(),
                  b1.data_ptr(), b2.data_ptr(), x.data_ptr(), xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None) == 0
    torch.cuda.synchronize()
    for w in (0, 1):
        t = trace[w * 2048:(w + 1) * 2048].cpu().numpy()
        n = int((t != 0).sum())
        t = t[:n] - t[0]
        fmt = lambda a: " ".join(f"{int(v):6d}" for v in a)
        print(f"{tag} wave {4 * w}: {n} stamps, total {int(t[-1])} ticks")
        print("  start pair     :", fmt(t[:2]))
        ps = t[2:2 + 48]
        print("  P steps pre    :", fmt(ps[0::2]))
        print("  P steps wait   :", fmt(ps[1::2] - ps[0::2]))
        print("  P steps work   :", fmt(ps[2::2] - ps[1:-1:2]))
        print("  ln2 begin/end, drain pair:", fmt(t[50:54]))
        print("  first FFN stamps:", fmt(t[54:62]))
