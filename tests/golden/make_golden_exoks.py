"""Generate the Ex-OKS golden fixtures (BASELINE config 5) from the REFERENCE's own evaluator.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_exoks.py

``exoks_cases.npz``: synthetic (image, category) cells - ground-truth instances with visibilities 0..3 (3 = outside
the activation window), detections with presence probabilities - and the similarity matrices the reference's
``COCOeval.computeExtendedOks`` (mmpose/evaluation/metrics/_cocoeval.py:540-707) returns for them, for several
``confidence_thr`` / ``padding`` / ``use_area`` settings and for ``original=True``.
``exoks_chain.npz``: CropCOCO-style crops end to end - the reference's ``generate_probmaps`` targets of jittered
ground truth, decoded by the reference's ``ProbMap.decode``, scored by the reference's Ex-OKS.
The fixtures are data only (inputs + expected outputs); no reference source is stored.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference, load_reference_eval  # noqa: E402

K = 17


def make_eval(ev, gts, dts, confidence_thr, padding, use_area, gt_visibilities):
    e = ev.COCOeval(None, None, "keypoints", use_area=use_area, extended_oks=True, confidence_thr=confidence_thr,
                    padding=padding)
    e._gts[(1, 1)] = gts
    e._dts[(1, 1)] = dts
    e.gt_visibilities = list(gt_visibilities)
    return e


def random_cell(rng, G, D, out_frac):
    """GT boxes in a 640x480 image, keypoints around them (some pushed outside -> v = 3, some unannotated -> v = 0),
    detections = jittered copies of random GTs with presence probabilities."""
    gts, dts = [], []
    for _ in range(G):
        w, h = rng.uniform(40, 200), rng.uniform(60, 300)
        x0, y0 = rng.uniform(0, 640 - w), rng.uniform(0, 480 - h)
        kp = np.zeros((K, 3))
        kp[:, 0] = rng.uniform(x0, x0 + w, K)
        kp[:, 1] = rng.uniform(y0, y0 + h, K)
        vis = rng.choice([0, 1, 2], K, p=[0.15, 0.25, 0.6])
        out = rng.random(K) < out_frac
        kp[out, 0] += rng.choice([-1, 1], out.sum()) * rng.uniform(0.8, 1.6, out.sum()) * w
        vis[out & (vis > 0)] = 3
        kp[:, 2] = vis
        kp[vis == 0, :2] = 0
        gts.append(dict(keypoints=kp.flatten().tolist(), bbox=[x0, y0, w, h], area=float(w * h * rng.uniform(0.3, 0.7)),
                        iscrowd=0))
    for _ in range(D):
        g = gts[rng.integers(0, G)]
        kp = np.array(g["keypoints"]).reshape(K, 3).copy()
        s = np.sqrt(g["bbox"][2] * g["bbox"][3])
        kp[:, :2] += rng.normal(0, 0.04 * s, (K, 2))
        kp[:, 2] = np.clip(rng.beta(2, 1.2, K) + rng.normal(0, 0.05, K), -0.1, 1.1)  # presence probability
        dts.append(dict(keypoints=kp.flatten().tolist(), score=float(rng.uniform(0.1, 1.0)),
                        visibilities=rng.uniform(0, 1, K).tolist()))
    return gts, dts


def finish_gts(gts, gt_visibilities):
    """What COCOeval._prepare leaves in gt['ignore'] for these synthetic instances (_cocoeval.py:303-362): per level
    True iff no keypoint of that level (k == 0)."""
    for g in gts:
        vis = np.array(g["keypoints"])[2::3]
        levels = [vis > 0] + [vis == v for v in gt_visibilities]
        g["ignore"] = [bool(m.sum() == 0) for m in levels]


def main():
    ev = load_reference_eval()
    rng = np.random.default_rng(20250930)
    out = {}
    settings = [(0.5, 1.25, True), (0.45, 1.25, True), (None, 1.25, True), (0.5, 1.0, False), (0.5, 1.5, True)]
    n = 0
    for G, D, out_frac in [(1, 1, 0.25), (2, 3, 0.2), (3, 4, 0.3), (1, 5, 0.0), (2, 2, 0.6), (4, 25, 0.2)]:
        gts, dts = random_cell(rng, G, D, out_frac)
        gt_vis = [1, 2, 3]
        finish_gts(gts, gt_vis)
        for thr, padding, use_area in settings:
            for original in (False, True):
                e = make_eval(ev, [dict(g) for g in gts], [dict(d) for d in dts], thr, padding, use_area, gt_vis)
                with contextlib.redirect_stdout(io.StringIO()):
                    ious = e.computeExtendedOks(1, 1, original=original)
                tag = f"c{n}"
                out[f"{tag}/gt_kpts"] = np.array([g["keypoints"] for g in gts], np.float64).reshape(G, K, 3)
                out[f"{tag}/gt_bbox"] = np.array([g["bbox"] for g in gts], np.float64)
                out[f"{tag}/gt_area"] = np.array([g["area"] for g in gts], np.float64)
                out[f"{tag}/gt_ignore"] = np.array([g["ignore"] for g in gts], np.uint8)
                out[f"{tag}/dt_kpts"] = np.array([d["keypoints"] for d in dts], np.float64).reshape(D, K, 3)
                out[f"{tag}/dt_score"] = np.array([d["score"] for d in dts], np.float64)
                out[f"{tag}/params"] = np.array([np.nan if thr is None else thr, padding, float(use_area), float(original)])
                out[f"{tag}/gt_visibilities"] = np.array(gt_vis, np.int32)
                out[f"{tag}/ious"] = np.stack([np.asarray(m, np.float64) for m in ious])  # (L+1, min(D, 20), G)
                n += 1
    out["sigmas"] = np.asarray(ev.COCOeval(None, None, "keypoints").sigmas, np.float64)
    out["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "exoks_cases.npz"), **out)

    # ---- end-to-end chain on CropCOCO-style crops (config 5): maps -> reference decode -> reference Ex-OKS
    ns = load_reference()
    codec = ns.KEYPOINT_CODECS.build(dict(type="ProbMap", input_size=(192, 256), heatmap_size=(48, 64), sigma=-1))
    chain = {}
    B = 6
    for b in range(B):
        # ground truth in input-pixel space (256x192 crop of a person box padded by 1.25): some keypoints outside
        kp = np.stack([rng.uniform(10, 182, K), rng.uniform(10, 246, K)], -1)
        vis = rng.choice([0, 1, 2], K, p=[0.1, 0.2, 0.7])
        outm = rng.random(K) < 0.2
        kp[outm, 0] = rng.choice([-1, 1], outm.sum()) * rng.uniform(30, 80, outm.sum()) + np.where(rng.random(outm.sum()) < 0.5, 0, 192)
        vis[outm & (vis > 0)] = 3
        jit = kp + rng.normal(0, 2.0, kp.shape)
        hm_kp = (jit / np.array([192, 256]) * np.array([47, 63]))[None]
        hm, _ = ns.utils.generate_probmaps((48, 64), hm_kp, np.ones((1, K), np.float32), sigma=-1)
        hm = np.maximum(hm - 0.3, 0).astype(np.float32)
        ssum = hm.reshape(K, -1).sum(-1)[:, None, None]
        hm = (hm / np.where(ssum > 0, ssum, 1)).astype(np.float32)
        prob = np.where(vis == 3, rng.beta(1.2, 4, K), rng.beta(5, 1.2, K)).astype(np.float32)  # presence probability head
        kpts, conf = codec.decode(hm)
        # the crop is the image here: bbox = the un-padded person box inside the 1.25x activation window
        bw, bh = 192 / 1.25, 256 / 1.25
        bbox = [(192 - bw) / 2, (256 - bh) / 2, bw, bh]
        g = dict(keypoints=np.concatenate([kp, vis[:, None]], -1).flatten().tolist(), bbox=bbox, area=float(bw * bh * 0.5), iscrowd=0)
        g["keypoints"] = [float(v) for v in g["keypoints"]]
        gkp = np.array(g["keypoints"]).reshape(K, 3)
        gkp[gkp[:, 2] == 0, :2] = 0
        g["keypoints"] = gkp.flatten().tolist()
        finish_gts([g], [1, 2, 3])
        d = dict(keypoints=np.concatenate([kpts[0], prob[:, None]], -1).flatten().tolist(), score=float(conf.mean()),
                 visibilities=prob.tolist())
        e = make_eval(ev, [g], [d], 0.5, 1.25, True, [1, 2, 3])
        with contextlib.redirect_stdout(io.StringIO()):
            ious = e.computeExtendedOks(1, 1)
        chain[f"b{b}/hm"] = hm
        chain[f"b{b}/prob"] = prob
        chain[f"b{b}/gt_kpts"] = gkp
        chain[f"b{b}/gt_bbox"] = np.array(bbox)
        chain[f"b{b}/gt_area"] = np.array(g["area"])
        chain[f"b{b}/gt_ignore"] = np.array(g["ignore"], np.uint8)
        chain[f"b{b}/ref_keypoints"] = kpts
        chain[f"b{b}/ref_conf"] = conf
        chain[f"b{b}/ref_ious"] = np.stack([np.asarray(m, np.float64) for m in ious])
    chain["B"] = np.array(B)
    np.savez_compressed(os.path.join(HERE, "exoks_chain.npz"), **chain)
    for f in ("exoks_cases.npz", "exoks_chain.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
