"""CPU restatement of the reference's Ex-mAP evaluator (TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this).

Follows ``COCOeval`` of mmpose/evaluation/metrics/_cocoeval.py for iouType "keypoints" and one category:
``_prepare`` (:161-422), ``evaluate`` (:424-503), ``evaluateImg`` (:709-887, the ``return_matching=False`` branch the
metric consumes), ``accumulate`` (:889-1009) and ``summarize``/``_summarizeKps`` (:1011-1061, :1136-1190); the
similarity itself is ``oracle/exoks_ref.py``. Annotation dicts in, the same ``eval`` arrays / ``stats`` out.
PINNED: tests/golden/exmap_cases.npz holds the reference's own precision / recall / scores / stats and per-image
matches for synthetic datasets (tests/golden/make_golden_exmap.py); tests/test_exmap.py checks this file against them.

Not restated (not consumed by ``CocoMetric.compute_metrics``, coco_metric.py:719-745): the extra
``return_matching=True`` pass that fills ``matched_pairs`` (:488-500).
"""
import numpy as np

from . import exoks_ref

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)  # Params.setKpParams :1249
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)  # :1250
AREA_RNG = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]  # :1252
AREA_LBL = ["all", "medium", "large"]
MAX_DETS = 20


def prepare(gts, dts, extended_oks, padding, ignore_near_bbox):
    """_cocoeval.py:161-422. Edits copies of the annotations: visibilities (border points, the {1, 2} restriction of the
    classic metric, v = 3 from ``pad_to_contain``), the per-level ``ignore`` list; drops detections without a positive
    confidence. Returns (gts, dts, gt_visibilities)."""
    gts = [dict(g) for g in gts]
    dts = [dict(d) for d in dts]
    levels = set()
    for g in gts:
        kp = np.array(g["keypoints"])
        vis = kp[2::3]
        if ignore_near_bbox:  # :228-246
            x0, y0, w, h = g["bbox"]
            x1, y1 = x0 + w, y0 + h
            tx, ty = 0.05 * w, 0.05 * h
            x, y = kp[0::3], kp[1::3]
            in_y = (y > y0 - ty) & (y < y1 + ty)
            in_x = (x > x0 - tx) & (x < x1 + tx)
            near = ((np.abs(x - x0) < tx) & in_y) | ((np.abs(x - x1) < tx) & in_y) | ((np.abs(y - y0) < ty) & in_x) | (
                (np.abs(y - y1) < ty) & in_x)
            vis[near] = 0
        if not extended_oks:  # :248-257
            vis[~((vis == 1) | (vis == 2))] = 0
        elif "pad_to_contain" in g:  # :262-271
            ptc = np.array(g["pad_to_contain"], dtype=np.float64)
            ptc[vis <= 0] = -1.0
            out = ptc > padding
            vis[(vis > 2) & (~out)] = 1
            vis[out] = 3
        levels.update(np.unique(vis.astype(int)).tolist())
        kp[2::3] = vis
        g["keypoints"] = kp.tolist()
        g["keypoints"][2::3] = vis.astype(int).tolist()  # :279
    gt_vis = [v for v in sorted(levels) if v > 0]
    for g in gts:  # :303-362: the final flags depend only on which visibility values the instance has
        vis = np.array(g["keypoints"])[2::3]
        present = np.unique(vis[vis > 0].astype(int))
        ign = np.ones(len(gt_vis) + 1, bool)
        ign[present] = False  # indexed by the visibility VALUE, as the reference does (:358)
        ign[0] = len(present) <= 0
        g["ignore"] = ign.tolist()
    kept = []
    for d in dts:  # :365-418
        conf = np.array(d["keypoints"])[2::3]
        if "visibilities" not in d:
            d["visibilities"] = conf
        if np.count_nonzero(conf > 0) == 0:
            continue
        kept.append(d)
    return gts, kept, gt_vis


def _arrays(gts, dts, n_levels, K):
    gk = np.array([g["keypoints"] for g in gts], np.float64).reshape(len(gts), K, 3)
    gb = np.array([g["bbox"] for g in gts], np.float64).reshape(len(gts), 4)
    ga = np.array([g.get("area", 0.0) for g in gts], np.float64)
    gi = np.array([g["ignore"] for g in gts], bool).reshape(len(gts), n_levels)
    dk = np.array([d["keypoints"] for d in dts], np.float64).reshape(len(dts), K, 3)
    ds = np.array([d["score"] for d in dts], np.float64)
    return gk, gb, ga, gi, dk, ds


def evaluate_img(gt, dt, ious, iou_i, a_rng, use_area, match_by_bbox, iou_thrs=IOU_THRS, max_det=MAX_DETS):
    """_cocoeval.py:709-887 for one image / level / area range. ``ious``: (D', G) of this level, detections in
    evaluation order, instances in annotation order (or an empty list). Returns None or a dict with dtMatches /
    gtMatches (ids, -1 = none), dtIgnore, gtIgnore, dtScores, gtIndices, and the similarities of the matches made."""
    if len(gt) == 0 and len(dt) == 0:
        return None
    flags = []
    for g in gt:
        area = g["area"] if ("area" in g and use_area) else g["bbox"][2] * g["bbox"][3] * 0.53
        flags.append(1 if (g["ignore"][iou_i] or area < a_rng[0] or area > a_rng[1]) else 0)
    gtind = np.argsort(flags, kind="mergesort")
    gt = [gt[i] for i in gtind]
    dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
    dt = [dt[i] for i in dtind[:max_det]]
    crowd = [int(g["iscrowd"]) for g in gt]
    iou = np.asarray(ious)[:, gtind] if len(ious) > 0 else ious
    T, G, D = len(iou_thrs), len(gt), len(dt)
    gtm = -np.ones((T, G), np.int64)
    dtm = -np.ones((T, D), np.int64)
    gt_ig = np.array([flags[i] for i in gtind])
    dt_ig = np.zeros((T, D))
    sims = []
    if len(iou):
        for ti, t in enumerate(iou_thrs):
            for di, d in enumerate(dt):
                best = min([t, 1 - 1e-10])
                m = -1
                if match_by_bbox:  # nearest box centre (L1, under 20 px) among the instances similar enough
                    nearest = 20
                    dc = np.array(d["bbox"][:2]) + np.array(d["bbox"][2:]) / 2
                    for gi, g in enumerate(gt):
                        if gtm[ti, gi] >= 0 and not crowd[gi]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if iou[di, gi] < t:
                            continue
                        gc = np.array(g["bbox"][:2]) + np.array(g["bbox"][2:]) / 2
                        dist = np.abs(dc - gc).sum()
                        if dist < nearest:
                            nearest, m, best = dist, gi, iou[di, gi]
                else:
                    for gi in range(G):
                        if gtm[ti, gi] >= 0 and not crowd[gi]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if iou[di, gi] < best:
                            continue
                        best, m = iou[di, gi], gi
                if m == -1:
                    continue
                sims.append(best)
                dt_ig[ti, di] = gt_ig[m]
                dtm[ti, di] = gt[m]["id"]
                gtm[ti, m] = d["id"]
    out_rng = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape(1, D)
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm < 0, np.repeat(out_rng, T, 0)))
    if np.all(gt_ig):  # also when the image has no instance at all
        dt_ig[:] = True
    return dict(dtIds=[d["id"] for d in dt], gtIds=[g["id"] for g in gt], dtMatches=dtm, gtMatches=gtm,
                dtScores=[d["score"] for d in dt], gtIgnore=gt_ig, dtIgnore=dt_ig, gtIndices=gtind, sims=sims)


def accumulate(eval_imgs, n_levels, n_areas, n_imgs, iou_thrs=IOU_THRS, rec_thrs=REC_THRS, max_det=MAX_DETS):
    """_cocoeval.py:889-1009. ``eval_imgs``: level-major, then area range, then image (one category).
    Returns precision (T, V, R, 1, A, 1), recall (T, V, 1, A, 1), scores like precision; -1 where nothing was evaluated."""
    T, R = len(iou_thrs), len(rec_thrs)
    precision = -np.ones((T, n_levels, R, 1, n_areas, 1))
    recall = -np.ones((T, n_levels, 1, n_areas, 1))
    scores = -np.ones((T, n_levels, R, 1, n_areas, 1))
    for v in range(n_levels):
        for a in range(n_areas):
            E = [e for e in eval_imgs[(v * n_areas + a) * n_imgs:(v * n_areas + a + 1) * n_imgs] if e is not None]
            if not E:
                continue
            sc = np.concatenate([e["dtScores"][:max_det] for e in E])
            order = np.argsort(-sc, kind="mergesort")
            sc_sorted = sc[order]
            dtm = np.concatenate([e["dtMatches"][:, :max_det] for e in E], axis=1)[:, order]
            dt_ig = np.concatenate([e["dtIgnore"][:, :max_det] for e in E], axis=1)[:, order]
            gt_ig = np.concatenate([e["gtIgnore"] for e in E])
            npig = np.count_nonzero(gt_ig == 0)
            if npig == 0:
                continue
            tps = np.logical_and(dtm >= 0, np.logical_not(dt_ig))
            fps = np.logical_and(dtm < 0, np.logical_not(dt_ig))
            tp_sum = np.cumsum(tps, axis=1).astype(np.float64)
            fp_sum = np.cumsum(fps, axis=1).astype(np.float64)
            for t in range(T):
                tp, fp = tp_sum[t], fp_sum[t]
                nd = len(tp)
                rc = tp / npig
                pr = tp / (fp + tp + np.spacing(1))
                recall[t, v, 0, a, 0] = rc[-1] if nd else 0
                pr = np.maximum.accumulate(pr[::-1])[::-1] if nd else pr  # the right-to-left "never decreasing" sweep
                q, ss = np.zeros(R), np.zeros(R)
                idx = np.searchsorted(rc, rec_thrs, side="left")
                ok = idx < nd  # thresholds the recall never reaches stay 0
                q[ok], ss[ok] = pr[idx[ok]], sc_sorted[idx[ok]]
                precision[t, v, :, 0, a, 0] = q
                scores[t, v, :, 0, a, 0] = ss
    return precision, recall, scores


def summarize(precision, recall, gt_vis, loc_similarities, iou_thrs=IOU_THRS):
    """_cocoeval.py:1017-1059 + :1136-1190: the 11 + len(gt_vis) numbers CocoMetric reports, with their names."""
    def mean_of(ap, iou_thr=None, area="all", visibility=None):
        a = AREA_LBL.index(area)
        v = 0 if visibility is None else gt_vis.index(visibility) + 1
        s = precision if ap else recall
        if iou_thr is not None:
            s = s[np.where(iou_thr == iou_thrs)[0]]
        s = s[:, v, :, :, a, 0] if ap else s[:, v, :, a, 0]
        s = s[s > -1]
        return -1 if len(s) == 0 else np.mean(s)

    stats = [mean_of(1)]
    names = ["AP"]
    for v in gt_vis:
        stats.append(mean_of(1, visibility=v))
        names.append("AP (v={:d})".format(v))
    for ap, tag in ((1, "AP"), (0, "AR")):
        stats += [mean_of(ap, iou_thr=0.5), mean_of(ap, iou_thr=0.75), mean_of(ap, area="medium"), mean_of(ap, area="large")]
        names += [tag + " .5", tag + " .75", tag + " (M)", tag + " (L)"]
        if ap:
            stats.append(mean_of(0))
            names.append("AR")
    stats.append(np.mean(loc_similarities))
    names.append("OKS")
    return np.array(stats, np.float64), names


def evaluate(gts, dts, sigmas, img_ids=None, use_area=True, extended_oks=True, match_by_bbox=False, confidence_thr=0.5,
             padding=1.25, ignore_near_bbox=False):
    """The whole ``evaluate(); accumulate(); summarize()`` sequence (coco_metric.py:720-722). Returns a dict with
    gt_visibilities, per-image results (level-major, then area range, then image), precision, recall, scores, stats,
    stats_names."""
    K = len(sigmas)
    gts, dts, gt_vis = prepare(gts, dts, extended_oks, padding, ignore_near_bbox)
    if img_ids is None:
        img_ids = sorted({g["image_id"] for g in gts})
    img_ids = list(np.unique(img_ids))
    L = len(gt_vis) + 1
    by_img_g = {i: [] for i in img_ids}
    by_img_d = {i: [] for i in img_ids}
    for g in gts:
        if g["image_id"] in by_img_g:
            by_img_g[g["image_id"]].append(g)
    for d in dts:
        if d["image_id"] in by_img_d:
            by_img_d[d["image_id"]].append(d)
    ious = {}
    for i in img_ids:
        g, d = by_img_g[i], by_img_d[i]
        if len(g) == 0 or len(d) == 0:
            ious[i] = [[] for _ in range(L)]
            continue
        gk, gb, ga, gi, dk, ds = _arrays(g, d, L, K)
        ious[i] = list(exoks_ref.extended_oks(gk, gb, ga, gi, dk, ds, sigmas, gt_vis, confidence_thr, padding, use_area,
                                              original=not extended_oks))
    eval_imgs = [evaluate_img(by_img_g[i], by_img_d[i], ious[i][v], v, a_rng, use_area, match_by_bbox)
                 for v in range(L) for a_rng in AREA_RNG for i in img_ids]
    sims = [s for e in eval_imgs if e is not None for s in e["sims"]]
    precision, recall, scores = accumulate(eval_imgs, L, len(AREA_RNG), len(img_ids))
    stats, names = summarize(precision, recall, gt_vis, sims)
    return dict(gt_visibilities=gt_vis, eval_imgs=eval_imgs, img_ids=img_ids, precision=precision, recall=recall,
                scores=scores, stats=stats, stats_names=names, loc_similarities=np.array(sims))
