"""Generate the Ex-mAP golden fixtures from the REFERENCE's own evaluator (build container only, needs /root/reference):

    python tests/golden/make_golden_exmap.py

``exmap_cases.npz``: synthetic keypoint datasets (ground-truth instances with visibilities 0..3, crowd instances,
``pad_to_contain`` values, images without instances / without detections, tied scores) and, for several evaluator
settings, what ``COCOeval.evaluate(); accumulate(); summarize()`` (mmpose/evaluation/metrics/_cocoeval.py:424-1190, the
sequence of coco_metric.py:720-722) produce for them: ``eval['precision'|'recall'|'scores']``, ``stats``, the mean
localisation similarity and the per-image matches. Data only (inputs + expected outputs); no reference source is stored.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference_eval  # noqa: E402

K = 17


class TinyCoco:
    """The four calls COCOeval makes on its cocoGt / cocoDt arguments."""

    def __init__(self, anns, img_ids):
        self.anns = anns
        self.img_ids = img_ids

    def getImgIds(self):
        return list(self.img_ids)

    def getCatIds(self):
        return [1]

    def getAnnIds(self, imgIds=(), catIds=()):
        keep = set(imgIds)
        return [i for i, a in enumerate(self.anns) if a["image_id"] in keep]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


def make_dataset(rng, n_img, with_ptc):
    gts, dts = [], []
    gid, did = 1, 1
    img_ids = list(range(100, 100 + n_img))
    for img in img_ids:
        mode = rng.integers(0, 10)
        G = 0 if mode == 0 else int(rng.integers(1, 5))
        D = 0 if mode == 1 else int(rng.integers(1, 8)) if mode < 8 else int(rng.integers(18, 27))
        if mode == 2:
            G, D = 0, 0
        here = []
        for _ in range(G):
            w, h = rng.uniform(30, 220), rng.uniform(40, 320)
            x0, y0 = rng.uniform(0, 640 - w), rng.uniform(0, 480 - h)
            kp = np.zeros((K, 3))
            kp[:, 0] = rng.uniform(x0, x0 + w, K)
            kp[:, 1] = rng.uniform(y0, y0 + h, K)
            vis = rng.choice([0, 1, 2], K, p=[0.2, 0.25, 0.55])
            out = rng.random(K) < 0.15
            kp[out, 0] += rng.choice([-1, 1], out.sum()) * rng.uniform(0.8, 1.6, out.sum()) * w
            vis[out & (vis > 0)] = 3
            if rng.random() < 0.1:
                vis[:] = 0  # an instance without annotated keypoints
            border = rng.random(K) < 0.1  # some points on the box edge (for ignore_near_bbox)
            kp[border, 0] = x0 + rng.uniform(-0.02, 0.02, border.sum()) * w
            kp[:, 2] = vis
            kp[vis == 0, :2] = 0
            g = dict(id=gid, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), bbox=[x0, y0, w, h],
                     area=float(w * h * rng.uniform(0.3, 0.7)), iscrowd=int(rng.random() < 0.12),
                     num_keypoints=int((vis > 0).sum()))
            if with_ptc:
                ptc = np.where(vis == 3, rng.uniform(1.0, 2.0, K), rng.uniform(0.3, 1.4, K))
                g["pad_to_contain"] = ptc.tolist()
            gts.append(g)
            here.append(g)
            gid += 1
        for j in range(D):
            if here and rng.random() < 0.8:
                g = here[rng.integers(0, len(here))]
                kp = np.array(g["keypoints"]).reshape(K, 3).copy()
                s = np.sqrt(g["bbox"][2] * g["bbox"][3])
                kp[:, :2] += rng.normal(0, rng.choice([0.005, 0.02, 0.06]) * s, (K, 2))
                src = g
            else:
                src = None
                kp = np.zeros((K, 3))
                kp[:, 0] = rng.uniform(0, 640, K)
                kp[:, 1] = rng.uniform(0, 480, K)
            gv = np.array(src["keypoints"])[2::3] if src is not None else np.ones(K)
            kp[:, 2] = np.clip(np.where(gv == 3, rng.beta(1.2, 4, K), rng.beta(5, 1.2, K)) + rng.normal(0, 0.05, K), -0.1, 1.1)
            if rng.random() < 0.03:
                kp[:, 2] = 0  # dropped by the evaluator (no positive confidence)
            x, y = kp[:, 0], kp[:, 1]
            score = float(np.round(rng.uniform(0.05, 1.0), 2 if rng.random() < 0.5 else 6))  # 2 decimals -> ties
            bbox = [float(x.min()), float(y.min()), float(x.max() - x.min()), float(y.max() - y.min())]
            if src is not None and rng.random() < 0.8:  # top-down: the detection carries (about) the box it was cropped from
                bbox = (np.array(src["bbox"]) + rng.normal(0, 2.0, 4)).tolist()
            dts.append(dict(id=did, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), score=score, bbox=bbox,
                            area=float(bbox[2] * bbox[3])))
            did += 1
    return gts, dts, img_ids


def main():
    ev = load_reference_eval()
    rng = np.random.default_rng(20251001)
    out = {}
    settings = [  # extended_oks, match_by_bbox, confidence_thr, padding, use_area, ignore_near_bbox, pad_to_contain, n_img
        (True, False, 0.5, 1.25, True, False, False, 40),
        (False, False, 0.5, 1.25, True, False, False, 40),
        (True, True, 0.5, 1.25, True, False, False, 30),
        (True, False, 0.4, 1.25, False, True, True, 30),
        (True, False, None, 1.5, True, False, True, 12),
    ]
    for n, (ext, mbb, thr, padding, use_area, near, with_ptc, n_img) in enumerate(settings):
        gts, dts, img_ids = make_dataset(rng, n_img, with_ptc)
        e = ev.COCOeval(TinyCoco(gts, img_ids), TinyCoco(dts, img_ids), "keypoints", sigmas=None, use_area=use_area,
                        extended_oks=ext, match_by_bbox=mbb, confidence_thr=thr, padding=padding, ignore_near_bbox=near)
        e.params.useSegm = None
        with contextlib.redirect_stdout(io.StringIO()):
            e.evaluate()
            e.accumulate()
            e.summarize()
        tag = f"s{n}"
        out[f"{tag}/settings"] = np.array([float(ext), float(mbb), np.nan if thr is None else thr, padding, float(use_area),
                                           float(near)])
        out[f"{tag}/img_ids"] = np.array(img_ids, np.int64)
        out[f"{tag}/gt_ids"] = np.array([[g["id"], g["image_id"], g["iscrowd"]] for g in gts], np.int64)
        out[f"{tag}/gt_kpts"] = np.array([g["keypoints"] for g in gts], np.float64)
        out[f"{tag}/gt_box"] = np.array([g["bbox"] + [g["area"]] for g in gts], np.float64)
        if with_ptc:
            out[f"{tag}/gt_ptc"] = np.array([g["pad_to_contain"] for g in gts], np.float64)
        out[f"{tag}/dt_ids"] = np.array([[d["id"], d["image_id"]] for d in dts], np.int64)
        out[f"{tag}/dt_kpts"] = np.array([d["keypoints"] for d in dts], np.float64)
        out[f"{tag}/dt_box"] = np.array([d["bbox"] + [d["area"], d["score"]] for d in dts], np.float64)
        out[f"{tag}/gt_visibilities"] = np.array(e.gt_visibilities, np.int64)
        out[f"{tag}/precision"] = e.eval["precision"]
        out[f"{tag}/recall"] = e.eval["recall"]
        out[f"{tag}/scores"] = e.eval["scores"]
        out[f"{tag}/stats"] = np.asarray(e.stats, np.float64)
        out[f"{tag}/stats_names"] = np.array(e.stats_names)
        out[f"{tag}/n_loc_similarities"] = np.array(len(e.loc_similarities))
        # per-image results, flattened: for every non-empty (level, area, image) the matched ids and ignore flags
        rows_dt, rows_gt = [], []
        for idx, r in enumerate(e.evalImgs):
            if r is None:
                continue
            T = r["dtMatches"].shape[0]
            for di, d_id in enumerate(r["dtIds"]):
                rows_dt.append([idx, d_id] + r["dtMatches"][:, di].tolist() + np.asarray(r["dtIgnore"])[:, di].astype(int).tolist())
            for gi, g_id in enumerate(r["gtIds"]):
                rows_gt.append([idx, g_id, int(r["gtIgnore"][gi])] + r["gtMatches"][:, gi].tolist())
            assert T == 10
        out[f"{tag}/img_dt_rows"] = np.array(rows_dt, np.int64).reshape(-1, 22)
        out[f"{tag}/img_gt_rows"] = np.array(rows_gt, np.int64).reshape(-1, 13)
        out[f"{tag}/img_none"] = np.array([r is None for r in e.evalImgs])
        # COCOeval.matched_pairs (_cocoeval.py:486-499): detection / instance pairs whose box centres coincide, with their
        # similarity at level 0 (nan for instances ignored there)
        out[f"{tag}/matched_pairs"] = np.array([[d["id"], g["id"], iou] for d, g, iou in e.matched_pairs], np.float64).reshape(-1, 3)
        print(tag, "gts", len(gts), "dts", len(dts), "levels", e.gt_visibilities, "AP", e.stats[0], "OKS", e.stats[-1])
    out["n_cases"] = np.array(len(settings))
    np.savez_compressed(os.path.join(HERE, "exmap_cases.npz"), **out)
    print("exmap_cases.npz", os.path.getsize(os.path.join(HERE, "exmap_cases.npz")))


if __name__ == "__main__":
    main()
