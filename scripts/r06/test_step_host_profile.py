#!/usr/bin/env python
"""Where does `model.test_step(batch)` spend its host time at bs 64? cProfile of 40 calls after warm-up (round 6: 76 % of the headline on one box,
86 % in round 5)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import apis  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

cfg = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
model = apis.init_model(cfg, {"state_dict": sd}, device="cuda:0")
B = 64
center, scale = S.whole_image_bbox_meta(B)
batches = [apis.pack_crops(S.synthetic_crops(B, seed=200 + i).cuda(), center, scale, model.dataset_meta) for i in range(3)]
with torch.no_grad():
    for i in range(6):
        model.test_step(batches[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        model.test_step(batches[i % 3])
    torch.cuda.synchronize()
    print(f"test_step: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per batch of {B}")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(40):
        model.test_step(batches[i % 3])
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
