#!/usr/bin/env python
"""Soak of the last-arriver LayerNorm tail (pp_skinny_linear): thousands of replays of the small-batch step at several batch sizes, two steps in
flight, EVERY result compared bit for bit with the first - a stale read of another XCD's rows would show as a transient mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import ProbPoseEngine  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402
from probpose_code_amd.dist import pack_records  # noqa: E402
from probpose_code_amd.pipeline import StepPipeline  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="f16x3")
bad = 0
for B in (1, 3, 8, 17):
    crops = [S.synthetic_crops(B, seed=900 + i).cuda() for i in range(2)]
    want = [pack_records(eng.forward(c, True, S.COCO_FLIP_INDICES)).cpu().numpy().copy() for c in crops]
    pipe = StepPipeline(eng, B, S.COCO_FLIP_INDICES, depth=2)
    prev = None
    for it in range(n_iter):
        t = pipe.submit(crops[it & 1])
        if prev is not None:
            got = pipe.result(prev[0])[0].numpy()
            if not np.array_equal(got, want[prev[1]]):
                bad += 1
                print(f"B {B} iteration {it - 1}: MISMATCH, max |diff| {np.nanmax(np.abs(got - want[prev[1]])):.3e}", flush=True)
        prev = (t, it & 1)
    got = pipe.result(prev[0])[0].numpy()
    bad += int(not np.array_equal(got, want[prev[1]]))
    print(f"B {B}: {n_iter} pipelined replays, mismatches so far {bad}", flush=True)
print("SOAK", "FAILED" if bad else "OK")
