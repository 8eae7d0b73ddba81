#!/usr/bin/env python
"""Which of the trained-like statistics costs the f16x3 path its margin? (round 6, GPU.) One statistic at a time switched OFF in
`synthetic_state_dict(stats="trained")`, 32 crops + flip, keypoint L_inf against oracle.model_ref.predict (fp32 CPU).

    python scripts/r06/trained_stats_ablation.py
Test/measurement infrastructure: imports oracle/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_ref as M  # noqa: E402
from probpose_code_amd import ProbPoseEngine  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count()))
crops = S.synthetic_crops(32, seed=100)
cases = [("unit", dict(stats="unit")), ("trained", {}), ("no small rows", dict(small_rows=1.0)), ("no massive", dict(massive=())),
         ("gamma one decade less", dict(gamma_decades=1.0)), ("no gamma spread", dict(gamma_decades=0.0)), ("no row offset", dict(row_offset=0.0)),
         ("small rows 1e-3", dict(small_rows=1e-3))]
for name, kw in cases:
    kw = dict(kw)
    stats = kw.pop("stats", "trained")
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0, stats=stats, **kw)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD)
    for prec in ("f16x3", "f32"):
        eng = ProbPoseEngine(sd, 12, precision=prec)
        out = eng.forward(crops.cuda(), True, S.COCO_FLIP_INDICES, return_features=True)
        torch.cuda.synchronize()
        d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
        from probpose_code_amd.weights import from_split
        feat = out["features"]
        feat = (from_split(feat.cpu()) if prec == "f16x3" else feat.float().cpu()).reshape(64, 16, 12, 384)[:32].permute(0, 3, 1, 2).numpy()
        fe = np.abs(feat - ref["features"]).max()
        pr = np.abs(out["scalars"][0].cpu().numpy()[:, None] - ref["keypoints_probs"]).max()
        print(f"{name:24s} {prec:6s} keypoints {d[d < 2].max():.2e} px  flips {int((d >= 2).sum())}  features {fe:.2e}  probs {pr:.1e}", flush=True)
