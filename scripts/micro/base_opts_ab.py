"""dev only: ViT-B 384x288 f16x3 step (bs 32, flip), per-tag kernel time under library options (same process):
python base_opts_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd import ProbPoseEngine, _lib
from probpose_code_amd import synthetic as S
B, img = 32, (384, 288)
sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
x = S.synthetic_crops(B, img_size=img, seed=1).cuda()
eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384))
for _ in range(2): eng.forward(x, True, S.COCO_FLIP_INDICES)
combos = [dict(), dict(linear_ovl=0), dict(psplit_nst=3), dict(psplit_nst=2), dict(linear_ovl=0, psplit_nst=3)]
defaults = {k: _lib.get_option(k) for k in ("linear_ovl", "psplit_nst")}
res = {}
for rep in range(2):
    for i, c in enumerate(combos):
        for k, v in defaults.items(): _lib.set_option(k, c.get(k, v))
        eng.forward(x, True, S.COCO_FLIP_INDICES)
        eng.profile = {}
        eng.forward(x, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        per = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in eng.profile.items()}
        eng.profile = None
        res.setdefault(i, []).append(per)
for k, v in defaults.items(): _lib.set_option(k, v)
for i, c in enumerate(combos):
    best = {k: min(r.get(k, 0.0) for r in res[i]) for k in res[i][0]}
    tot = sum(best.values())
    print(c or "defaults", f"total {tot:.2f} ms:", {k: round(v, 2) for k, v in sorted(best.items(), key=lambda kv: -kv[1])[:6]})
