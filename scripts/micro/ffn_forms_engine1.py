"""dev: the first fused projection + FFN launch inside the engine, both forms: inputs and outputs snapshotted around the call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L, synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
from probpose_code_amd.weights import from_split

eng = ProbPoseEngine(S.synthetic_state_dict("small", seed=0, logit_scale=2.0), 12, precision="f16x3", device="cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
crops = S.synthetic_crops(B, seed=5).cuda()
ws = eng._workspace(B, 2, 0)
orig = eng._call
snaps = {}
for form in (0, 1):
    L.set_option("ffn_dma_waves", form)
    seen = []
    def call(tag, name, *args, _form=form):
        first = tag == "proj_ffn_split" and not seen
        if first:
            torch.cuda.synchronize()
            pre = {k: ws[k].clone() for k in ("x", "att", "h", "hs")}
        r = orig(tag, name, *args)
        if first:
            torch.cuda.synchronize()
            seen.append(1)
            snaps[_form] = (pre, {k: ws[k].clone() for k in ("x", "att", "h", "hs")}, args)
        return r
    eng._call = call
    eng.run_backbone(crops, True)
    torch.cuda.synchronize()
eng._call = orig
L.set_option("ffn_dma_waves", 1)
dec = lambda k, t: t.float() if k == "x" else from_split(t).float()
for k in ("x", "att", "h", "hs"):
    a0, a1 = dec(k, snaps[0][0][k]), dec(k, snaps[1][0][k])
    print(f"before the call, {k}: forms equal {torch.equal(snaps[0][0][k].view(torch.int32), snaps[1][0][k].view(torch.int32))}, NaN {torch.isnan(a0).sum().item()} / {torch.isnan(a1).sum().item()}, |max| {a0.abs().max().item():.3e}")
for k in ("x", "h", "hs"):
    a0, a1 = dec(k, snaps[0][1][k]), dec(k, snaps[1][1][k])
    d = (a0 - a1).abs()
    print(f"after the call, {k}: NaN {torch.isnan(a0).sum().item()} / {torch.isnan(a1).sum().item()}, max |diff| {d[~torch.isnan(d)].max().item() if (~torch.isnan(d)).any() else float('nan'):.3e}, first NaN rows (form 1) {torch.isnan(a1).any(1).nonzero().flatten()[:12].tolist()}")
print("args equal:", [a == b for a, b in zip(snaps[0][2], snaps[1][2])])
h0, h1 = snaps[0][1]["h"].view(torch.int32), snaps[1][1]["h"].view(torch.int32)
ne = (h0 != h1)
far = ((h0 - h1).abs() > 8) & ne
print(f"h words differing: {ne.sum().item()}, by more than rounding: {far.sum().item()}; rows {far.any(1).nonzero().flatten()[:40].tolist()}; words of the first such row: {far[far.any(1).nonzero().flatten()[0]].nonzero().flatten()[:40].tolist() if far.any() else []}")
rows = far.any(1).nonzero().flatten()
print("block rows (mod 96) histogram:", torch.bincount(rows % 96, minlength=96).tolist())
print("word (mod 16) histogram:", torch.bincount(far.nonzero()[:, 1] % 16, minlength=16).tolist())
