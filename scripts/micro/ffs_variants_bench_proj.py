"""dev only: time pp_proj_ffn_split_residual_layernorm of the libraries built by ffs_variants.sh, round-robin minima
(python ffs_variants_bench_proj.py tag1 tag2 ...); CHECK=1 compares outputs with the first tag's (bitwise)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = int(os.environ.get("M", 24576)), 384, 1536
torch.manual_seed(0)
att = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
wp = to_split(torch.randn(E, E) / E ** 0.5).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
bp, b1, b2, g, be = torch.randn(E).cuda() * 0.1, torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
P = ctypes.c_void_p
libs = {}
for tag in dict.fromkeys(sys.argv[1:]):
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
    libs[tag] = (lib, packed, wpp, fn)
xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda"); hs = torch.empty(M, E, device="cuda")
def run(tag):
    lib, packed, wpp, fn = libs[tag]
    assert fn(att.data_ptr(), wpp.data_ptr(), bp.data_ptr(), g.data_ptr(), be.data_ptr(), hs.data_ptr(), packed.data_ptr(),
              b1.data_ptr(), b2.data_ptr(), x.data_ptr(), xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None) == 0
for _ in range(200): run(next(iter(libs)))
torch.cuda.synchronize()
times = {t: [] for t in libs}
for rep in range(int(os.environ.get("REPS", 6))):
    for tag in libs:
        for _ in range(3): run(tag)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(tag)
        e1.record(); torch.cuda.synchronize()
        times[tag].append(e0.elapsed_time(e1) / 20 * 1e3)
ref = None
for tag in libs:
    ts = sorted(times[tag]); msg = ""
    if os.environ.get("CHECK"):
        run(tag); torch.cuda.synchronize()
        cur = (xo.clone(), ho.clone())
        if ref is None: ref = cur
        else: msg = "  same as first: %s" % (torch.equal(cur[0], ref[0]) and torch.equal(cur[1].view(torch.int32), ref[1].view(torch.int32)))
    print(f"{tag:>14}: min {ts[0]:7.1f} median {ts[len(ts) // 2]:7.1f} us{msg}", flush=True)
