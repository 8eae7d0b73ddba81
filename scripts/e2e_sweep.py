import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from probpose_code_amd import ProbPoseEngine, synthetic as S
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
e16 = ProbPoseEngine(sd, 12, precision="bf16"); e32 = ProbPoseEngine(sd, 12, precision="f32")
for B in [1, 2, 3, 5, 7, 33, 64, 65, 100, 130]:
    x = S.synthetic_crops(B, seed=B).cuda()
    a = e16.forward(x, True, S.COCO_FLIP_INDICES); b = e32.forward(x, True, S.COCO_FLIP_INDICES)
    ka, kb = a["keypoints"].cpu().numpy(), b["keypoints"].cpu().numpy()
    d = np.abs(ka - kb).max(-1)
    same = d < 2.0
    print(B, "nan", bool(np.isnan(ka).any()), "agree", round(float(same.mean()), 3), "max diff on agreeing", round(float(d[same].max()), 3),
          "probs diff", round(float((a["scalars"][0] - b["scalars"][0]).abs().max()), 4))
    # graph path equals eager
    if B in (3, 64):
        e16.capture(B, True, S.COCO_FLIP_INDICES).copy_(x)
        g = e16.forward_graph(x, True, S.COCO_FLIP_INDICES)
        print("   graph == eager:", bool(torch.equal(g["keypoints"], a["keypoints"])))
