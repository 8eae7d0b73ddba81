#!/usr/bin/env python
"""Chaos soak of StepPipeline with HOST (pinned) batches of random size, depth 2 and 3, graph replay for full batches: every record compared bit for bit
with the kernel-by-kernel result of the same crops.   python scripts/r06/chaos_pipeline_host.py [seconds]"""
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
from probpose_code_amd.dist import pack_records  # noqa: E402
from probpose_code_amd.pipeline import StepPipeline  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="f16x3")
fi = S.COCO_FLIP_INDICES
MAXB = 32
sizes = [1, 2, 3, 5, 8, 13, 17, 18, 21, 32, 32, 32]
host = {(B, v): S.synthetic_crops(B, seed=900 + 7 * B + v).pin_memory() for B in set(sizes) for v in range(2)}
want = {k: pack_records(eng.forward(c.cuda(), True, fi)).cpu().numpy().copy() for k, c in host.items()}
rng = random.Random(11)
n, bad = 0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    depth = rng.choice([1, 2, 3])
    pipe = StepPipeline(eng, MAXB, fi, flip_test=True, depth=depth, use_graph="full")
    keys = [(rng.choice(sizes), rng.randint(0, 1)) for _ in range(rng.randint(3, 40))]
    pending = []
    for key in keys:
        while len(pending) >= depth:
            t, k = pending.pop(0)
            got = pipe.result(t)[0, :k[0]].numpy()
            n += 1
            if not np.array_equal(got, want[k]):
                bad += 1
                print(f"MISMATCH depth {depth} B {k[0]}: max |diff| {np.nanmax(np.abs(got - want[k])):.3e}", flush=True)
        src = host[key] if rng.random() < 0.7 else host[key].cuda()
        pending.append((pipe.submit(src), key))
    for t, k in pending:
        got = pipe.result(t)[0, :k[0]].numpy()
        n += 1
        if not np.array_equal(got, want[k]):
            bad += 1
            print(f"MISMATCH (drain) depth {depth} B {k[0]}", flush=True)
print(f"{n} batches in {seconds:.0f} s, {bad} mismatches, graph captures {eng.graph_captures}")
print("PIPELINE SOAK", "FAILED" if bad else "OK")
