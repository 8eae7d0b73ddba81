"""Golden vectors for the OKS suppression of the metric driver, from the REFERENCE's own
``mmpose/evaluation/functional/nms.py`` (``oks_iou`` :58-116, ``oks_nms`` :119-170, ``soft_oks_nms`` :173-259). Build container only:

    python tests/golden/make_golden_nms.py

``nms_cases.npz``: per case the instances of one image (keypoints (N, 17, 3), score, area), the threshold, the kept
indices and the OKS of instance 0 to the others. Data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference_nms  # noqa: E402


def main():
    nms = load_reference_nms()
    rng = np.random.default_rng(20251002)
    out = {}
    cases = [(1, 0.9), (2, 0.9), (6, 0.9), (12, 0.5), (12, 0.9), (25, 0.7)]
    for n, (N, thr) in enumerate(cases):
        base = rng.uniform(50, 400, (max(N // 3, 1), 17, 2))
        kp = np.zeros((N, 17, 3))
        for i in range(N):
            kp[i, :, :2] = base[rng.integers(0, len(base))] + rng.normal(0, rng.choice([1.0, 6.0, 30.0]), (17, 2))
            kp[i, :, 2] = rng.uniform(0, 1, 17)
        score = np.round(rng.uniform(0.1, 1.0, N), 3)
        area = rng.uniform(3000, 30000, N)
        db = [dict(keypoints=kp[i], score=score[i], area=area[i]) for i in range(N)]
        keep = np.asarray(nms.oks_nms(db, thr, sigmas=None), np.int64)
        out[f"n{n}/kpts"], out[f"n{n}/score"], out[f"n{n}/area"], out[f"n{n}/thr"] = kp, score, area, np.array(thr)
        out[f"n{n}/keep"] = keep
        out[f"n{n}/soft_keep"] = np.asarray(nms.soft_oks_nms([dict(d) for d in db], thr, sigmas=None), np.int64)
        out[f"n{n}/soft_keep_max5"] = np.asarray(nms.soft_oks_nms([dict(d) for d in db], thr, max_dets=5, sigmas=None), np.int64)
        out[f"n{n}/iou0"] = nms.oks_iou(kp[0].flatten(), kp[1:].reshape(N - 1, -1), area[0], area[1:]) if N > 1 else np.zeros(0, np.float32)
        out[f"n{n}/iou0_vis"] = nms.oks_iou(kp[0].flatten(), kp[1:].reshape(N - 1, -1), area[0], area[1:], vis_thr=0.4) if N > 1 else np.zeros(0, np.float32)
    # box NMS of the multi-person demo (nms.py:16-55)
    for n, (N, thr) in enumerate([(0, 0.3), (1, 0.3), (9, 0.3), (30, 0.5), (30, 0.1)]):
        xy = rng.uniform(0, 400, (N, 2))
        wh = rng.uniform(20, 200, (N, 2))
        dets = np.concatenate([xy, xy + wh, np.round(rng.uniform(0.05, 1.0, (N, 1)), 2)], 1)
        if N >= 9:
            dets[1, :4] = dets[0, :4] + 3.0   # near-duplicates and a tied score
            dets[2, 4] = dets[3, 4]
        out[f"box{n}/dets"], out[f"box{n}/thr"] = dets, np.array(thr)
        out[f"box{n}/keep"] = np.asarray(nms.nms(dets, thr), np.int64)
    out["n_box_cases"] = np.array(5)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **out)
    print("nms_cases.npz", os.path.getsize(os.path.join(HERE, "nms_cases.npz")), [out[f"n{n}/keep"].tolist() for n in range(len(cases))])


if __name__ == "__main__":
    main()
