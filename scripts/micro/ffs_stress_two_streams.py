"""dev only: pp_proj_ffn_split_residual_layernorm on two streams at once (two independent problems), outputs compared with
the solo results: python ffs_stress_two_streams.py tag ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd.weights import to_split
here = os.path.dirname(os.path.abspath(__file__))
M, E, Fd = 24576, 384, 1536
P = ctypes.c_void_p
def problem(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    d = dict(att=to_split(r(M, E)).cuda(), x=r(M, E).cuda(), wp=to_split(r(E, E) / E ** 0.5).cuda(), w1=to_split(r(Fd, E) / E ** 0.5).cuda(),
             w2=to_split(r(E, Fd) / Fd ** 0.5).cuda(), bp=(r(E) * 0.1).cuda(), b1=(r(Fd) * 0.1).cuda(), b2=(r(E) * 0.1).cuda(),
             g=torch.ones(E).cuda(), be=torch.zeros(E).cuda())
    d.update(xo=torch.empty(M, E, device="cuda"), ho=torch.empty(M, E, device="cuda"), hs=torch.empty(M, E, device="cuda"))
    return d
for tag in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(here, "build", f"libffs_{tag}.so"))
    lib.pp_ffn_split_packed_bytes.restype = ctypes.c_longlong
    pk = lib.pp_ffn_split_pack_weights; pk.restype = ctypes.c_int; pk.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P]
    pp_ = lib.pp_proj_split_pack_weights; pp_.restype = ctypes.c_int; pp_.argtypes = [P, P, ctypes.c_int, P]
    fn = lib.pp_proj_ffn_split_residual_layernorm; fn.restype = ctypes.c_int
    fn.argtypes = [P] * 13 + [ctypes.c_float, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    probs = [problem(1), problem(2)]
    for d in probs:
        d["packed"] = torch.empty(lib.pp_ffn_split_packed_bytes(E, Fd) // 4, device="cuda")
        assert pk(d["w1"].data_ptr(), d["w2"].data_ptr(), d["packed"].data_ptr(), E, Fd, None) == 0
        d["wpp"] = torch.empty(E * E, device="cuda")
        assert pp_(d["wp"].data_ptr(), d["wpp"].data_ptr(), E, None) == 0
    def run(d, stream):
        st = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        assert fn(d["att"].data_ptr(), d["wpp"].data_ptr(), d["bp"].data_ptr(), d["g"].data_ptr(), d["be"].data_ptr(), d["hs"].data_ptr(),
                  d["packed"].data_ptr(), d["b1"].data_ptr(), d["b2"].data_ptr(), d["x"].data_ptr(), d["xo"].data_ptr(), d["g"].data_ptr(),
                  d["be"].data_ptr(), 1e-6, d["ho"].data_ptr(), M, E, Fd, st) == 0
    want = []
    for d in probs:
        run(d, None); torch.cuda.synchronize()
        want.append((d["xo"].clone(), d["ho"].clone()))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for it in range(40):
        for d in probs: d["xo"].fill_(float("nan")); d["ho"].fill_(float("nan")); d["hs"].fill_(float("nan"))
        torch.cuda.synchronize()
        for rep in range(3):
            run(probs[0], s0); run(probs[1], s1)
        torch.cuda.synchronize()
        for d, (xo, ho) in zip(probs, want):
            if not torch.equal(d["xo"], xo):
                rows = (d["xo"] != xo).any(1).nonzero().flatten()
                bad += 1
                if bad <= 4: print(f"  {tag} it {it}: {rows.numel()} rows differ, tiles {sorted(set((rows // 96).tolist()))[:12]}")
    print(f"{tag}: {bad} of 80 comparisons differ")
