"""dev only: time pp_qkv_attention_split of the libraries built by qka_variants.sh, round-robin minima (python qka_variants_bench.py tag ...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
n_seq, S, E, H, hd = 128, 192, 384, 12, 32
M = n_seq * S
torch.manual_seed(0)
h = to_split(torch.randn(M, E)).cuda(); w = to_split(torch.randn(3 * E, E) / E ** 0.5).cuda(); b = torch.randn(3 * E).cuda() * 0.1
out = torch.empty(M, E, device="cuda")
P = ctypes.c_void_p
libs = {}
for tag in dict.fromkeys(sys.argv[1:]):
    lib = ctypes.CDLL(os.path.join(here, "build", f"libqka_{tag}.so"))
    fn = lib.pp_qkv_attention_split; fn.restype = ctypes.c_int
    fn.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, P]
    libs[tag] = fn
def run(tag):
    assert libs[tag](h.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None) == 0
for _ in range(200): run(next(iter(libs)))
torch.cuda.synchronize()
times = {t: [] for t in libs}
for rep in range(6):
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
        run(tag); torch.cuda.synchronize(); cur = out.clone()
        if ref is None: ref = cur
        else: msg = "  same as first: %s" % torch.equal(cur.view(torch.int32), ref.view(torch.int32))
    print(f"{tag:>12}: min {ts[0]:7.1f} median {ts[len(ts) // 2]:7.1f} us{msg}", flush=True)
