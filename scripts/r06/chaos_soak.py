#!/usr/bin/env python
"""Chaos soak of the drop-in estimator: minutes of `model.test_step` and `test_step_stream(depth 2)` on batches of RANDOM size (1 .. 40 crops: the
small-batch plan, its boundary at 17 / 18 crops, the row-owner plan; eager launches, graph captures, evictions of the graph cache), every result
compared bit for bit with the first result the same batch produced - a race in the LayerNorm tails, a stale graph or a recycled workspace shows as
a mismatch.   python scripts/r06/chaos_soak.py [seconds]"""
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

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
MODE = os.environ.get("CHAOS_MODE", "both")      # step | stream | both
VERBOSE = os.environ.get("CHAOS_VERBOSE", "0") == "1"
cfg = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
model = apis.init_model(cfg, {"state_dict": sd}, device="cuda:0")
if os.environ.get("CHAOS_HEATMAPS") == "1":  # (test_step only: the stream API does not carry heatmaps)
    model.test_cfg["output_heatmaps"] = True
    MODE = "step"
if os.environ.get("CHAOS_ONE_HEAD_STREAM") == "1":
    model.engine.plan["head_two_streams"] = False
DEPTH = int(os.environ.get("CHAOS_DEPTH", "2"))
rng = random.Random(1234)
sizes = [1, 1, 1, 2, 2, 3, 4, 5, 6, 8, 8, 12, 16, 17, 18, 19, 24, 32, 40]
batches = {}
for B in sorted(set(sizes)):
    for v in range(2):
        center, scale = S.whole_image_bbox_meta(B)
        batches[(B, v)] = (S.synthetic_crops(B, seed=7000 + 10 * B + v).cuda(), center, scale)


def make(key):
    crops, center, scale = batches[key]
    return apis.pack_crops(crops, center, scale, model.dataset_meta)


def signature(samples):
    parts = []
    for s in samples:
        parts += [s.pred_instances.keypoints.ravel(), s.pred_instances.keypoint_scores.ravel(), s.pred_instances.keypoints_visible.ravel()]
        if "pred_fields" in s and "heatmaps" in s.pred_fields:
            hm = s.pred_fields.heatmaps
            parts.append(np.asarray(hm.cpu() if hasattr(hm, "cpu") else hm, dtype=np.float64).ravel())
    return np.concatenate(parts)


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20


want, n_calls, n_bad = {}, 0, 0
mem_marks = []
t_start = time.time()
t_end = t_start + seconds
with torch.no_grad():
    while time.time() < t_end:
        if len(mem_marks) < 4 and time.time() - t_start >= (len(mem_marks) + 1) * seconds / 4.2:
            mem_marks.append((round(torch.cuda.memory_reserved() / 2 ** 20), round(rss_mb())))
        if MODE == "step" or (MODE == "both" and rng.random() < 0.5):  # a burst of single calls
            for _ in range(rng.randint(1, 12)):
                key = (rng.choice(sizes), rng.randint(0, 1))
                if VERBOSE:
                    print("step", key, "graphs", len(model.engine._graphs), flush=True)
                sig = signature(model.test_step(make(key)))
                n_calls += 1
                if key not in want:
                    want[key] = sig
                elif not np.array_equal(sig, want[key]):
                    n_bad += 1
                    print(f"MISMATCH test_step B {key[0]} variant {key[1]}: max |diff| {np.nanmax(np.abs(sig - want[key])):.3e}", flush=True)
        else:  # a stream of batches, two in flight
            keys = [(rng.choice(sizes), rng.randint(0, 1)) for _ in range(rng.randint(2, 16))]
            if VERBOSE:
                print("stream", keys, flush=True)
            for key, samples in zip(keys, model.test_step_stream((make(k) for k in keys), depth=DEPTH)):
                sig = signature(samples)
                n_calls += 1
                if key not in want:
                    want[key] = sig
                elif not np.array_equal(sig, want[key]):
                    n_bad += 1
                    print(f"MISMATCH test_step_stream B {key[0]} variant {key[1]}: max |diff| {np.nanmax(np.abs(sig - want[key])):.3e}", flush=True)
eng = getattr(model, "engine", None)
print(f"{n_calls} batches in {seconds:.0f} s, {len(want)} distinct, {n_bad} mismatches" + (f", graph captures {eng.graph_captures}" if eng is not None and hasattr(eng, "graph_captures") else ""))
print("device MiB reserved / host RSS MiB at the quarter marks:", mem_marks)
print("CHAOS SOAK", "FAILED" if n_bad else "OK")
