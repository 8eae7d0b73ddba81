"""dev only: time the libraries built by ffs_variants.sh (python ffs_variants_bench.py tag1 tag2 ...); CHECK=1 also compares
x_out / h_out with the first tag's outputs (bitwise)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = int(os.environ.get("M", 24576)), 384, 1536
torch.manual_seed(0)
h = to_split(torch.randn(M, E)).cuda(); x = torch.randn(M, E).cuda()
w1 = to_split(torch.randn(Fd, E) / E ** 0.5).cuda(); w2 = to_split(torch.randn(E, Fd) / Fd ** 0.5).cuda()
b1, b2, g, be = torch.randn(Fd).cuda() * 0.1, torch.randn(E).cuda() * 0.1, torch.ones(E).cuda(), torch.zeros(E).cuda()
P = ctypes.c_void_p
ref = None
libs = {}
for tag in dict.fromkeys(sys.argv[1:]):
    lib = ctypes.CDLL(os.path.join(here, "build", f"libffs_{tag}.so"))
    lib.pp_ffn_split_packed_bytes.restype = ctypes.c_longlong
    packed = torch.empty(lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
    pk = lib.pp_ffn_split_pack_weights; pk.restype = ctypes.c_int; pk.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P]
    assert pk(w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, Fd, None) == 0
    fn = lib.pp_ffn_split_residual_layernorm; fn.restype = ctypes.c_int
    fn.argtypes = [P] * 8 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    libs[tag] = (lib, packed, fn)
xo = torch.empty(M, E, device="cuda"); ho = torch.empty(M, E, device="cuda")
def run(tag):
    lib, packed, fn = libs[tag]
    st = fn(h.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(), xo.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6, ho.data_ptr(), M, E, Fd, None)
    assert st == 0
# warm the chip up on the first variant, then round-robin over the variants (a position in the order is worth +-8 % otherwise)
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
for tag in libs:
    ts = sorted(times[tag]); us = ts[0]; med = ts[len(ts) // 2]
    msg = ""
    if os.environ.get("CHECK"):
        run(tag); torch.cuda.synchronize()
        cur = (xo.clone(), ho.clone())
        if ref is None: ref = cur
        else: msg = "  same as first: %s" % (torch.equal(cur[0], ref[0]) and torch.equal(cur[1].view(torch.int32), ref[1].view(torch.int32)))
    print(f"{tag:>14}: min {us:7.1f} median {med:7.1f} us  {12 * M * E * Fd / us / 1e6:6.0f} TF of MFMA issue{msg}", flush=True)
