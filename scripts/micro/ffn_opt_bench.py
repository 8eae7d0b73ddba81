"""dev: the fused f16x3 feed-forward launches at bs 64 (M = 24 576, F = 1536) under the values of a library option, round-robin minima over
hipEvent-timed bursts:   python scripts/micro/ffn_opt_bench.py ffn_skew "0 1" [tag]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split

opt, vals = (sys.argv[1], [int(v) for v in sys.argv[2].split()]) if len(sys.argv) > 2 else ("ffn_skew", [1])
tag = sys.argv[3] if len(sys.argv) > 3 else ""
M, E, F_ = 24576, 384, 1536
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
h, r, att = to_split(rnd(M, E)).cuda(), rnd(M, E).cuda(), to_split(rnd(M, E)).cuda()
w1, w2, wp = to_split(rnd(F_, E, scale=E ** -0.5)).cuda(), to_split(rnd(E, F_, scale=F_ ** -0.5)).cuda(), to_split(rnd(E, E, scale=E ** -0.5)).cuda()
vec = [rnd(n, scale=0.2).cuda() for n in (E, E, E, F_, E, E, E)]
packed = torch.empty(L.lib.pp_ffn_split_packed_bytes(E, F_) // 4, device="cuda")
L.call("pp_ffn_split_pack_weights", w1.data_ptr(), w2.data_ptr(), packed.data_ptr(), E, F_, None)
wpp = torch.empty(E * E, device="cuda")
L.call("pp_proj_split_pack_weights", wp.data_ptr(), wpp.data_ptr(), E, None)
xo, ho, hs = torch.empty(M, E, device="cuda"), torch.empty(M, E, device="cuda"), torch.empty(M, E, device="cuda")

def ffn():
    L.call("pp_ffn_split_residual_layernorm", h.data_ptr(), packed.data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), r.data_ptr(), xo.data_ptr(),
           vec[5].data_ptr(), vec[6].data_ptr(), 1e-6, ho.data_ptr(), M, E, F_, None)

def proj():
    L.call("pp_proj_ffn_split_residual_layernorm", att.data_ptr(), wpp.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), hs.data_ptr(),
           packed.data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), r.data_ptr(), xo.data_ptr(), vec[5].data_ptr(), vec[6].data_ptr(), 1e-6, ho.data_ptr(), M, E, F_, None)

best = {}
for rep in range(6):
    for v in vals:
        L.set_option(opt, v)
        for name, fn in (("ffn", ffn), ("proj_ffn", proj)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(24):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 24 * 1e3
            best[(name, v)] = min(best.get((name, v), 1e9), t)
print(f"{tag:>12s} " + "   ".join(f"{name} {opt}={v}: {best[(name, v)]:.1f} us" for name in ("ffn", "proj_ffn") for v in vals))
