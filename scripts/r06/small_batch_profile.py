#!/usr/bin/env python
"""Where does a small batch spend its step? (round 6, item 2.) Per launch tag: HIP-event time of the eager launches (engine.profile), and the
replayed hipGraph's step time one step in flight, for B in --batches.

    python scripts/r06/small_batch_profile.py [--batches 1,8] [--arch small]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import ProbPoseEngine  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,8")
ap.add_argument("--plan", default="")
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
plan = {}
for kv in filter(None, args.plan.split(",")):
    k, v = kv.split("=")
    plan[k] = {"0": False, "1": True}.get(v, v)
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="f16x3", plan=plan or None)
print("layer plan:", eng.layer_plan, getattr(eng, "small_plan", None))
for B in [int(b) for b in args.batches.split(",")]:
    crops = S.synthetic_crops(B, seed=1).cuda()
    for _ in range(3):
        eng.forward(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    eng.profile = {}
    for _ in range(5):
        eng.forward(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    prof, eng.profile = eng.profile, None
    tot = 0.0
    rows = []
    for tag, evs in prof.items():
        t = sum(a.elapsed_time(b) for a, b in evs) / 5 * 1e3
        rows.append((t, tag, len(evs) // 5))
        tot += t
    print(f"--- B = {B}: eager kernel time per step {tot:.0f} us over {sum(r[2] for r in rows)} launches")
    for t, tag, n in sorted(rows, reverse=True):
        print(f"    {tag:18s} {n:3d} launches {t:8.1f} us  ({t / n:6.1f} each)")
    eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print(f"    hipGraph replay, one step in flight: {dt * 1e3:.3f} ms per step = {B / dt:.0f} crops/s")
