"""dev only: per-kernel time of the ViT-B 384x288 f16x3 step's tower convolutions under the K-walk options (same process,
round-robin): python base_conv_ab.py"""
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
res = {}
for rep in range(3):
    for ti in (0, 1, 2):
        _lib.set_option("psplit_tap_inner", ti)
        eng.forward(x, True, S.COCO_FLIP_INDICES)
        eng.profile = {}
        eng.forward(x, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        per = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in eng.profile.items()}
        eng.profile = None
        for k in ("conv3x3", "deconv", "deconv_head"):
            res.setdefault((ti, k), []).append(per.get(k, 0.0))
_lib.set_option("psplit_tap_inner", 1)
for (ti, k), v in sorted(res.items()): print(f"tap_inner={ti} {k:12s} min {min(v):.3f} ms")
