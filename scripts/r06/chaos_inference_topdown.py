#!/usr/bin/env python
"""Chaos soak of the caller-level flow: `inference_topdown(model, img, boxes)` against `inference_topdown_stream` over frames of random size with 0 .. 24
random boxes (boxes partly outside the image, slivers, one-pixel boxes, xywh / xyxy): equal results frame by frame, finite keypoints.
   python scripts/r06/chaos_inference_topdown.py [seconds]"""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import apis  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
model = apis.init_model(cfg, {"state_dict": sd}, device="cuda:0")
rng = random.Random(99)
nrng = np.random.default_rng(5)


def frame():
    h, w = rng.choice([(480, 640), (720, 1280), (333, 517), (64, 48), (1080, 1920)])
    img = nrng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    n = rng.choice([0, 0, 1, 1, 2, 3, 5, 8, 13, 24])
    boxes = []
    for _ in range(n):
        kind = rng.random()
        if kind < 0.6:
            x0, y0 = rng.uniform(0, w * 0.8), rng.uniform(0, h * 0.8)
            bw, bh = rng.uniform(4, w * 0.6), rng.uniform(4, h * 0.9)
        elif kind < 0.8:  # partly / mostly outside
            x0, y0 = rng.uniform(-w * 0.5, w), rng.uniform(-h * 0.5, h)
            bw, bh = rng.uniform(10, w), rng.uniform(10, h)
        elif kind < 0.9:  # sliver
            x0, y0, bw, bh = rng.uniform(0, w - 2), rng.uniform(0, h - 2), rng.uniform(0.5, 2), rng.uniform(20, h)
        else:  # one pixel
            x0, y0, bw, bh = rng.uniform(0, w - 1), rng.uniform(0, h - 1), 1.0, 1.0
        boxes.append([x0, y0, x0 + bw, y0 + bh])
    return img, (np.asarray(boxes, dtype=np.float32) if boxes else None)


def sig(samples):
    return np.concatenate([np.concatenate([s.pred_instances.keypoints.ravel(), s.pred_instances.keypoint_scores.ravel()]) for s in samples])


n, bad, nonfinite = 0, 0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    frames = [frame() for _ in range(rng.randint(2, 12))]
    single = [sig(apis.inference_topdown(model, img, bb)) for img, bb in frames]
    for a, out in zip(single, apis.inference_topdown_stream(model, frames, depth=2, max_persons=64)):
        b = sig(out)
        n += 1
        if not np.isfinite(a).all():
            nonfinite += 1
        if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
            bad += 1
            print(f"MISMATCH frame of {len(out)} boxes", flush=True)
print(f"{n} frames in {seconds:.0f} s, {bad} mismatches, {nonfinite} frames with non-finite keypoints")
print("TOPDOWN SOAK", "FAILED" if bad else "OK")
