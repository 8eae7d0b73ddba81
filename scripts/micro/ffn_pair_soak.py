# dev only: soak of the paired-chunk FFN kernels (in-place inline-assembly MFMAs, hand-placed settle points): the two fused entry points on two
# streams at once, thousands of launches, every result compared bit for bit with the solo launch
import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
L = T._lib()
M, E, F_ = 24576, 384, 1536
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
probs = []
for seed in (100, 200):
    h, r, w1, b1, w2, b2, g, be = T._ffn_inputs(M, F_, seed=seed)
    att, wp, bp, g2, be2 = T._proj_inputs(M, seed=seed + 20)
    wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
    wps = T._sp(wp)
    L.call("pp_proj_split_pack_weights", wps.data_ptr(), wpp.data_ptr(), E, None)
    probs.append(dict(att=T._sp(att), h=T._sp(h), x=r.cuda(), wpp=wpp, packed=T._ffn_pack(L, w1, w2, E, F_), dev=[t.cuda() for t in (bp, g2, be2, b1, b2, g, be)],
                      xo=torch.empty(M, E, device="cuda"), ho=torch.empty(M, E, device="cuda"), hs=torch.empty(M, E, device="cuda")))

def proj(d, s):
    v = d["dev"]
    L.call("pp_proj_ffn_split_residual_layernorm", d["att"].data_ptr(), d["wpp"].data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), d["hs"].data_ptr(),
           d["packed"].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), d["x"].data_ptr(), d["xo"].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), 1e-6, d["ho"].data_ptr(), M, E, F_,
           None if s is None else s.cuda_stream)

def ffn(d, s):
    v = d["dev"]
    L.call("pp_ffn_split_residual_layernorm", d["h"].data_ptr(), d["packed"].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), d["x"].data_ptr(), d["xo"].data_ptr(),
           v[5].data_ptr(), v[6].data_ptr(), 1e-6, d["ho"].data_ptr(), M, E, F_, None if s is None else s.cuda_stream)

assert L.get_option("ffn_pair") == 1
want = []
for d, fn in zip(probs, (proj, ffn)):
    fn(d, None); torch.cuda.synchronize()
    want.append((d["xo"].clone(), d["ho"].clone()))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
bad, t0 = 0, time.time()
for it in range(iters):
    for d in probs:
        d["xo"].fill_(float("nan")); d["ho"].fill_(float("nan"))
    torch.cuda.synchronize()
    for _ in range(5):
        proj(probs[0], s0); ffn(probs[1], s1)
    torch.cuda.synchronize()
    for k, (d, (xo, ho)) in enumerate(zip(probs, want)):
        if not (torch.equal(d["xo"], xo) and torch.equal(d["ho"].view(torch.int32), ho.view(torch.int32))):
            bad += 1
            print("MISMATCH iteration", it, "stream", k, (d["xo"] != xo).sum().item(), "elements", flush=True)
print(f"{iters} iterations x 5 launches x 2 streams in {time.time() - t0:.1f} s: {bad} mismatching results")
sys.exit(1 if bad else 0)
