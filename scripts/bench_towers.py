"""dev only: the small tower stages of the f16x3 step at bs 64 (split-K convolutions + sum-pool + tower_final), per kernel tag,
with the library's slice rule and with the channel-range slices switched off (option ksplit_channels = 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from probpose_code_amd import synthetic as S, _lib
from probpose_code_amd.engine import ProbPoseEngine
B = 64
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
crops = S.synthetic_crops(B, seed=100).cuda()
for ch in (1, 0, 1, 0):
    _lib.set_option("ksplit_channels", ch)
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    for _ in range(3): eng.forward(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    eng.profile = {}
    for _ in range(10): eng.forward(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    prof = {k: float(np.sum([a.elapsed_time(b) for a, b in v])) / 10 for k, v in eng.profile.items()}
    eng.profile = None
    per = {k: [round(a.elapsed_time(b) * 1e3, 1) for a, b in v[-2:]] for k, v in [("conv3x3_splitk", [])]}
    print(f"ksplit_channels={ch}: slices {[eng.ksplit(128, 4, 4), eng.ksplit(128, 2, 2)]}  conv3x3_splitk {prof['conv3x3_splitk'] * 1e3:.1f} us  maxpool {prof['maxpool'] * 1e3:.1f} us  "
          f"conv3x3 {prof['conv3x3'] * 1e3:.1f}  step sum {sum(prof.values()):.3f} ms")
    del eng
