import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from probpose_code_amd import ProbPoseEngine
from probpose_code_amd import synthetic as S
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
for name, below in (("small", 100000), ("row_owner", 0)):
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    eng.small_rows_below = below
    for B in (12, 16, 20, 24, 28, 32):
        crops = S.synthetic_crops(B, seed=1).cuda()
        for _ in range(3): eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f"{name:10s} B {B:3d}: {dt * 1e3:.3f} ms", flush=True)
