"""CPU restatement of the reference's Ex-OKS similarity (TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this).

Follows ``COCOeval.computeExtendedOks`` (mmpose/evaluation/metrics/_cocoeval.py:540-707) for iouType "keypoints" and
``fix_bbox_aspect_ratio`` (mmpose/structures/keypoint/keypoints_min_padding.py:68-133), on arrays instead of annotation
dicts. PINNED: tests/golden/exoks_cases.npz holds the reference's own outputs for 60 synthetic cells
(tests/golden/make_golden_exoks.py); tests/test_exoks.py checks this file against them to 1e-12.
"""
import numpy as np

MAX_DETS = 20  # Params.setKpParams: maxDets = [20] (_cocoeval.py:1246-1256)


def fix_bbox_aspect_ratio_xyxy(bb_xyxy, padding, aspect_ratio=3 / 4):
    """keypoints_min_padding.py:68-133 for one xyxy box: centre kept, the short side grown to the 3:4 aspect ratio, both
    sides scaled by `padding`. The new sizes pass through float32 exactly as the reference's ``astype(np.float32)``."""
    x0, y0, x1, y1 = (float(v) for v in bb_xyxy)
    cx, cy = x0 + (x1 - x0) / 2, y0 + (y1 - y0) / 2
    w, h = x1 - x0, y1 - y0
    nw, nh = np.float32(w), np.float32(h)
    if w == 0:
        w = 1.0
    if h == 0:
        h = 1.0
    if w / h > aspect_ratio:
        nh = np.float32(w / aspect_ratio)
    else:
        nw = np.float32(h * aspect_ratio)
    nw = np.float32(nw * np.float32(padding))
    nh = np.float32(nh * np.float32(padding))
    # float64 centre +- float32 size / 2  (numpy promotes float64 op float32 -> float64)
    return cx - np.float64(nw / np.float32(2)), cy - np.float64(nh / np.float32(2)), cx + np.float64(nw / np.float32(2)), cy + np.float64(nh / np.float32(2))


def sort_detections(scores, max_dets=MAX_DETS):
    """Stable descending order by score, truncated (``np.argsort(-score, kind='mergesort')``, _cocoeval.py:546-550)."""
    order = np.argsort(-np.asarray(scores, np.float64), kind="mergesort")
    return order[:max_dets]


def extended_oks(gt_kpts, gt_bbox, gt_area, gt_ignore, dt_kpts, dt_score, sigmas, gt_visibilities, confidence_thr=0.5,
                 padding=1.25, use_area=True, original=False):
    """gt_kpts (G,K,3) [x,y,v], gt_bbox (G,4) xywh, gt_area (G,), gt_ignore (G,L+1) bool, dt_kpts (D,K,3) [x,y,presence],
    dt_score (D,). Returns (L+1, min(D, 20), G) float64: level 0 is v > 0, level l is v == gt_visibilities[l-1]."""
    gt_kpts = np.asarray(gt_kpts, np.float64)
    dt_kpts = np.asarray(dt_kpts, np.float64)
    order = sort_detections(dt_score)
    dt_kpts = dt_kpts[order]
    G, D = gt_kpts.shape[0], dt_kpts.shape[0]
    L = len(gt_visibilities) + 1
    k = len(sigmas)
    var = (np.asarray(sigmas, np.float64) * 2) ** 2
    out = np.zeros((L, D, G))
    for j in range(G):
        xg, yg, vg = gt_kpts[j, :, 0], gt_kpts[j, :, 1], gt_kpts[j, :, 2]
        gt_in_img = vg < 3
        masks = [vg > 0] + [vg == v for v in gt_visibilities]
        bb = gt_bbox[j]
        if original:
            x0, x1, y0, y1 = bb[0] - bb[2], bb[0] + bb[2] * 2, bb[1] - bb[3], bb[1] + bb[3] * 2
        else:
            x0, y0, x1, y1 = fix_bbox_aspect_ratio_xyxy([bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]], padding)
        area = gt_area[j] if use_area else bb[3] * bb[2] * 0.53
        for i in range(D):
            xd, yd = dt_kpts[i, :, 0], dt_kpts[i, :, 1]
            cd = np.clip(dt_kpts[i, :, 2], 0, 1)
            if confidence_thr is not None:
                cd = (cd >= confidence_thr).astype(int)
            for lvl, m in enumerate(masks):
                k1 = int(np.count_nonzero(m))
                assert not (gt_ignore[j][lvl] and k1 > 0)
                if k1 > 0:
                    dist = (xd - xg) ** 2 + (yd - yg) ** 2
                    if not original:
                        e_pred = np.minimum(xd - x0, x1 - xd) ** 2 + np.minimum(yd - y0, y1 - yd) ** 2
                        e_gt = np.minimum(xg - x0, x1 - xg) ** 2 + np.minimum(yg - y0, y1 - yg) ** 2
                        dist = dist.copy()
                        sel = ~gt_in_img & (cd == 1)
                        dist[sel] = e_pred[sel]
                        sel = gt_in_img & (cd == 0)
                        dist[sel] = e_gt[sel]
                        sel = ~gt_in_img & (cd == 0)
                        dist[sel] = 0
                else:
                    z = np.zeros(k)
                    dx = np.maximum(z, x0 - xd) + np.maximum(z, xd - x1)
                    dy = np.maximum(z, y0 - yd) + np.maximum(z, yd - y1)
                    dist = dx ** 2 + dy ** 2
                e = dist / var / (area + np.spacing(1)) / 2
                if k1 > 0:
                    e = e[m]
                out[lvl, i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return out
