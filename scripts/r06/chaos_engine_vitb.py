#!/usr/bin/env python
"""Chaos soak at engine level for ViT-B 384x288 (BASELINE config 4 geometry) and for the bf16 / f32 modes of ViT-S: random batch sizes, graph cache of
three (evictions and re-captures), every result compared bit for bit with the kernel-by-kernel launch.   python chaos_engine_vitb.py [seconds]"""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import ProbPoseEngine  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
fi = S.COCO_FLIP_INDICES
cases = [("ViT-B 384x288 f16x3", dict(arch="base", img=(384, 288), prec="f16x3", sizes=(1, 2, 3, 5, 8, 12, 16, 33))),
         ("ViT-S bf16", dict(arch="small", img=(256, 192), prec="bf16", sizes=(1, 2, 7, 17, 18, 40, 64))),
         ("ViT-S f32", dict(arch="small", img=(256, 192), prec="f32", sizes=(1, 3, 8, 20)))]
ok = True
for name, c in cases:
    sd = S.synthetic_state_dict(c["arch"], img_size=c["img"], seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, img_size=c["img"], precision=c["prec"], input_size=(c["img"][1], c["img"][0]))
    eng.max_graphs = 3
    crops = {B: S.synthetic_crops(B, img_size=c["img"], seed=300 + B).cuda() for B in c["sizes"]}
    want = {}
    for B, x in crops.items():
        o = eng.forward(x, True, fi)
        want[B] = (o["keypoints"].cpu().numpy().copy(), o["scalars"].cpu().numpy().copy())
    rng = random.Random(2)
    n, bad = 0, 0
    t_end = time.time() + seconds / len(cases)
    while time.time() < t_end:
        B = rng.choice(c["sizes"])
        o = eng.forward_graph(crops[B], True, fi) if rng.random() < 0.8 else eng.forward(crops[B], True, fi)
        n += 1
        if not (np.array_equal(o["keypoints"].cpu().numpy(), want[B][0]) and np.array_equal(o["scalars"].cpu().numpy(), want[B][1])):
            bad += 1
            print(f"MISMATCH {name} B {B}", flush=True)
    ok &= bad == 0
    print(f"{name}: {n} steps, {bad} mismatches, {eng.graph_captures} captures", flush=True)
    del eng
    torch.cuda.empty_cache()
print("ENGINE SOAK", "OK" if ok else "FAILED")
