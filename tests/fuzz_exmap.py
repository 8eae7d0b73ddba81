#!/usr/bin/env python
"""Differential fuzz of the HIP Ex-mAP evaluator (probpose_code_amd.evaluation.COCOeval) against the oracle (oracle/exmap_ref.py - test infrastructure)
over random datasets and every switch combination: image counts 1 .. 120, empty images, crowds, score ties, zero-area boxes, keypoints outside the
box, all-invisible annotations. Exact equality of precision / recall / scores tables and stats.   python tests/fuzz_exmap.py [seconds]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import exmap_ref  # noqa: E402
from probpose_code_amd.evaluation import COCOeval  # noqa: E402

K = 17
SIGMAS = np.array([0.26, 0.25, 0.25, 0.35, 0.35, 0.79, 0.79, 0.72, 0.72, 0.62, 0.62, 1.07, 1.07, 0.87, 0.87, 0.89, 0.89]) / 10.0
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0


def dataset(rng):
    gts, dts = [], []
    n_img = int(rng.integers(1, 121))
    for img in range(n_img):
        here = []
        for _ in range(int(rng.integers(0, 7)) if rng.random() < 0.85 else 0):
            w, h = rng.uniform(1, 260), rng.uniform(1, 340)
            if rng.random() < 0.03:
                w = 0.0
            x0, y0 = rng.uniform(-20, 640), rng.uniform(-20, 480)
            kp = np.zeros((K, 3))
            spread = rng.choice([1.0, 1.0, 1.6])  # some keypoints outside the box
            kp[:, 0] = rng.uniform(x0 - (spread - 1) * w, x0 + spread * w + 1e-3, K)
            kp[:, 1] = rng.uniform(y0 - (spread - 1) * h, y0 + spread * h + 1e-3, K)
            vis = rng.choice([0, 1, 2, 3], K, p=[0.2, 0.2, 0.45, 0.15])
            if rng.random() < 0.05:
                vis[:] = 0
            kp[:, 2] = vis
            kp[vis == 0, :2] = 0
            g = dict(id=len(gts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), bbox=[x0, y0, w, h],
                     area=float(w * h * rng.choice([0.5, 1.0, 0.0 if rng.random() < 0.1 else 0.3])), iscrowd=int(rng.random() < 0.1))
            gts.append(g)
            here.append(g)
        for _ in range(int(rng.integers(0, 26))):
            if here and rng.random() < 0.8:
                g = here[rng.integers(0, len(here))]
                kp = np.array(g["keypoints"]).reshape(K, 3).copy()
                kp[:, :2] += rng.normal(0, rng.choice([0.0, 0.005, 0.02, 0.08]) * np.sqrt(abs(g["bbox"][2] * g["bbox"][3]) + 1.0), (K, 2))
                kp[:, 2] = np.where(kp[:, 2] == 3, rng.beta(1.2, 4, K), rng.beta(5, 1.2, K))
                bbox = list(g["bbox"])
            else:
                kp = np.stack([rng.uniform(0, 640, K), rng.uniform(0, 480, K), rng.uniform(0, 1, K)], 1)
                bbox = [float(kp[:, 0].min()), float(kp[:, 1].min()), float(np.ptp(kp[:, 0])), float(np.ptp(kp[:, 1]))]
            dts.append(dict(id=len(dts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(),
                            score=float(np.round(rng.uniform(0.05, 1.0), rng.choice([1, 2, 6]))), bbox=bbox, area=float(bbox[2] * bbox[3])))
    return gts, dts, list(range(n_img + int(rng.integers(0, 4))))


n, bad = 0, 0
t_end = time.time() + seconds
seed = 0
while time.time() < t_end:
    rng = np.random.default_rng(1000 + seed)
    seed += 1
    gts, dts, img_ids = dataset(rng)
    if not gts:
        continue
    for ext, mbb, use_area, near in itertools.product((True, False), (False, True), (True, False), (False, True)):
        kw = dict(use_area=use_area, extended_oks=ext, match_by_bbox=mbb, confidence_thr=float(rng.choice([0.3, 0.5, 0.7])),
                  padding=float(rng.choice([1.0, 1.25, 1.5])), ignore_near_bbox=near)
        try:
            ref = exmap_ref.evaluate(gts, dts, SIGMAS, img_ids=img_ids, **kw)
            e = COCOeval(gts, dts, "keypoints", sigmas=SIGMAS, **kw)
            e.params.imgIds = img_ids
            e.evaluate()
            e.accumulate()
            e.summarize()
            ok = (np.array_equal(e.eval["precision"], ref["precision"]) and np.array_equal(e.eval["recall"], ref["recall"])
                  and np.array_equal(e.eval["scores"], ref["scores"]) and np.array_equal(e.stats[:-1], ref["stats"][:-1])
                  and abs(e.stats[-1] - ref["stats"][-1]) <= 1e-12 and e.stats_names == ref["stats_names"])
        except Exception as exc:  # both raising the same way is agreement; report anything else
            ok = False
            print(f"seed {seed - 1} {kw}: {type(exc).__name__}: {exc}", flush=True)
        n += 1
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed - 1} {kw} (gts {len(gts)}, dts {len(dts)})", flush=True)
print(f"{n} evaluations over {seed} datasets in {seconds:.0f} s, {bad} mismatches")
print("EXMAP FUZZ", "FAILED" if bad else "OK")
