"""Ex-OKS on the device (BASELINE config 5).

``extended_oks`` mirrors ``COCOeval.computeExtendedOks`` (mmpose/evaluation/metrics/_cocoeval.py:540-707) for one
(image, category) cell on arrays: it orders the detections the way the evaluator does (stable, descending score, at
most ``maxDets`` = 20), launches ``pp_extended_oks`` and returns the (levels, detections, instances) similarity tensor
on the device. The matching / accumulation that follows in the reference (``evaluateImg``, ``accumulate``) consumes
exactly this tensor.
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

MAX_DETS = 20  # Params.setKpParams (_cocoeval.py:1246-1256)

COCO_SIGMAS = np.array([0.26, 0.25, 0.25, 0.35, 0.35, 0.79, 0.79, 0.72, 0.72, 0.62, 0.62, 1.07, 1.07, 0.87, 0.87, 0.89,
                        0.89]) / 10.0  # _cocoeval.py:107-130


def _dev64(x, device):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, np.float64))
    return t.to(device=device, dtype=torch.float64).contiguous()


def extended_oks(gt_kpts, gt_bbox, gt_area, gt_ignore, dt_kpts, dt_score, sigmas=COCO_SIGMAS,
                 gt_visibilities: Sequence[int] = (1, 2, 3), confidence_thr: Optional[float] = 0.5, padding: float = 1.25,
                 use_area: bool = True, original: bool = False, device="cuda", max_dets: int = MAX_DETS) -> torch.Tensor:
    """gt_kpts (G, K, 3) [x, y, v], gt_bbox (G, 4) xywh, gt_area (G,), gt_ignore (G, L + 1) bool, dt_kpts (D, K, 3)
    [x, y, presence probability], dt_score (D,). Arrays or tensors (host or device). Returns a float64 device tensor
    (L + 1, min(D, max_dets), G); level 0 is v > 0, level l is v == gt_visibilities[l - 1]. Raises like the reference:
    AssertionError for padding < 1 and for an instance flagged ignore at a level where it has keypoints."""
    assert padding >= 1.0, "Padding must be greater than or equal to 1.0"  # _cocoeval.py:560
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("probpose_code_amd.evaluation.extended_oks runs on the GPU only (no CPU fallback)")
    gk, gb, ga = _dev64(gt_kpts, device), _dev64(gt_bbox, device), _dev64(gt_area, device)
    dk, ds = _dev64(dt_kpts, device), _dev64(dt_score, device)
    G, D = gk.shape[0], dk.shape[0]
    L = len(gt_visibilities) + 1
    if G == 0 or D == 0:
        return torch.zeros((L, min(D, max_dets), G), dtype=torch.float64, device=device)
    K = gk.shape[1]
    vis = gk[:, :, 2]
    counts = torch.stack([(vis > 0).sum(1)] + [(vis == v).sum(1) for v in gt_visibilities], 1)  # (G, L)
    ign = torch.as_tensor(np.asarray(gt_ignore.cpu() if isinstance(gt_ignore, torch.Tensor) else gt_ignore, bool)).to(device)
    assert not bool((ign & (counts > 0)).any()), "k1 is negative but gt is not ignored"  # _cocoeval.py:654
    order = torch.sort(-ds, stable=True).indices[:max_dets]  # np.argsort(-score, kind="mergesort")
    dk = dk[order].contiguous()
    D = dk.shape[0]
    sg = _dev64(sigmas, device)
    gv = torch.as_tensor(list(gt_visibilities), dtype=torch.int32, device=device)
    out = torch.empty((L, D, G), dtype=torch.float64, device=device)
    thr = float("nan") if confidence_thr is None else float(confidence_thr)
    _lib.call("pp_extended_oks", gk.data_ptr(), gb.data_ptr(), ga.data_ptr(), dk.data_ptr(), sg.data_ptr(), gv.data_ptr(),
              G, D, K, L - 1, thr, float(padding), int(use_area), int(original), out.data_ptr(),
              torch.cuda.current_stream(device).cuda_stream)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The whole evaluator: evaluate() -> accumulate() -> summarize(), the sequence CocoMetric runs (coco_metric.py:720-722)
# ---------------------------------------------------------------------------------------------------------------------
class Params:
    """Params.setKpParams (_cocoeval.py:1245-1256)."""

    def __init__(self, iouType="keypoints"):
        if "keypoints" not in iouType:
            raise Exception("iouType not supported")  # the box / mask metrics are not on this path
        self.imgIds = []
        self.catIds = []
        self.iouThrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = [MAX_DETS]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ["all", "medium", "large"]
        self.useCats = 1
        self.iouType = iouType
        self.useSegm = None


def _annotations(src, img_ids):
    """Annotation dicts of a COCO-API object (getAnnIds / loadAnns, as xtcocotools.coco.COCO) or of a plain list."""
    if hasattr(src, "loadAnns"):
        return src.loadAnns(src.getAnnIds(imgIds=img_ids))
    keep = set(img_ids)
    return [a for a in src if a["image_id"] in keep]


class COCOeval:
    """Ex-OKS / OKS keypoint evaluation with the interface of the reference's ``COCOeval``
    (mmpose/evaluation/metrics/_cocoeval.py:23-160) for iouType "keypoints" and one category: same constructor
    arguments, ``params``, ``evaluate()``, ``accumulate()``, ``summarize()``, and afterwards ``eval`` (precision / recall /
    scores / counts), ``stats``, ``stats_names``, ``gt_visibilities``.

    ``cocoGt`` / ``cocoDt``: COCO-API objects (``getImgIds`` / ``getAnnIds`` / ``loadAnns``) or plain lists of annotation
    dicts (then ``params.imgIds`` defaults to the image ids of the ground truth). The annotation dicts are parsed on the
    host like the reference does (``_prepare``); similarities, matching and the precision / recall tables run on the GPU
    (pp_exoks_cells, pp_exoks_match, pp_exmap_accumulate); the last step averages the small tables on the host.
    Not provided: ``matched_pairs`` (the extra bbox-matching pass, :488-500, which CocoMetric does not read), the
    wholebody / crowd iouTypes."""

    def __init__(self, cocoGt=None, cocoDt=None, iouType="keypoints", sigmas=None, use_area=True, extended_oks=False,
                 match_by_bbox=False, confidence_thr=0.5, padding=1.25, ignore_near_bbox=False, device="cuda"):
        if iouType != "keypoints":
            raise Exception("iouType not supported")
        self.sigmas = np.asarray(sigmas, np.float64) if sigmas is not None else COCO_SIGMAS
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Params(iouType)
        if cocoGt is not None:
            if hasattr(cocoGt, "getImgIds"):
                self.params.imgIds = sorted(cocoGt.getImgIds())
                self.params.catIds = sorted(cocoGt.getCatIds())
            else:
                self.params.imgIds = sorted({g["image_id"] for g in cocoGt})
                self.params.catIds = sorted({g.get("category_id", 1) for g in cocoGt}) or [1]
        self.use_area = use_area
        self.score_key = "score"
        self.extended_oks = extended_oks
        self.confidence_thr = confidence_thr
        self.match_by_bbox = match_by_bbox
        self.padding = padding
        self.ignore_near_bbox = ignore_near_bbox
        self.device = torch.device(device)
        self.eval, self.stats, self.stats_names = {}, [], []
        self.gt_visibilities = []
        self.loc_similarities = []
        self._d = None

    # ---- host: annotation dicts -> flat arrays (COCOeval._prepare, _cocoeval.py:161-422)
    def _prepare(self):
        p = self.params
        K = len(self.sigmas)
        gts = list(_annotations(self.cocoGt, p.imgIds))  # (read only: edited visibilities / flags live in arrays)
        dts = list(_annotations(self.cocoDt, p.imgIds))
        # the reference forms (image, category) cells and loads annotations by catIds (:170-176, :415-422); this evaluator
        # forms cells by image only, which is the same thing for ONE category (COCO person) and nothing else
        cats = {a.get("category_id", 1) for a in gts} | {a.get("category_id", 1) for a in dts}
        if len(cats) > 1:
            raise NotImplementedError(f"COCOeval (MI355X) evaluates one category at a time; annotations carry {sorted(cats)}")
        if self.use_area:  # `tmparea = gt["area"]` raises KeyError in the reference (:696-697); never guess an area
            for g in gts:
                if "area" not in g:
                    raise KeyError("area")
        # instances: all keypoints as one array, the visibility edits of the reference vectorised over the instances
        gkp = np.array([g["keypoints"] for g in gts], np.float64).reshape(len(gts), K, 3)
        gbb = np.array([g["bbox"] for g in gts], np.float64).reshape(len(gts), 4)
        vis = gkp[:, :, 2].copy()
        if self.ignore_near_bbox:  # keypoints within 5 % of the box edge are not evaluated (:228-246)
            x0, y0, w, h = (gbb[:, i:i + 1] for i in range(4))
            x1, y1, tx, ty = x0 + w, y0 + h, 0.05 * w, 0.05 * h
            x, y = gkp[:, :, 0], gkp[:, :, 1]
            in_y, in_x = (y > y0 - ty) & (y < y1 + ty), (x > x0 - tx) & (x < x1 + tx)
            vis[((np.abs(x - x0) < tx) | (np.abs(x - x1) < tx)) & in_y | ((np.abs(y - y0) < ty) | (np.abs(y - y1) < ty)) & in_x] = 0
        if not self.extended_oks:  # the classic metric knows only v in {1, 2} (:248-257)
            vis[~((vis == 1) | (vis == 2))] = 0
        else:  # v = 3 <=> the keypoint needs more padding than the activation window has (:262-271)
            has = np.array(["pad_to_contain" in g for g in gts], bool)
            if has.any():
                ptc = np.array([g["pad_to_contain"] for g, hh in zip(gts, has) if hh], np.float64).reshape(-1, K)
                v = vis[has]
                ptc[v <= 0] = -1.0
                out = ptc > self.padding
                v[(v > 2) & (~out)] = 1
                v[out] = 3
                vis[has] = v
        vis = vis.astype(int)  # (:273: truncation, as the reference's astype)
        gkp[:, :, 2] = vis
        self.gt_visibilities = [int(v) for v in np.unique(vis) if v > 0]
        L = len(self.gt_visibilities) + 1
        # per-level ignore flags (:303-362): level index = the visibility VALUE (:358); level 0 = no annotated keypoint
        g_ignore = np.ones((len(gts), L), bool)
        if len(gts):
            for v in np.unique(vis[vis > 0]):
                g_ignore[(vis == v).any(1), v] = False  # (IndexError for a value beyond the level count, as in the reference)
            g_ignore[:, 0] = ~(vis > 0).any(1)
        dkp = np.array([d["keypoints"] for d in dts], np.float64).reshape(len(dts), K, 3)
        keep = (dkp[:, :, 2] > 0).any(1)  # detections without a positive confidence are dropped (:412-416)
        dts = [d for d, k in zip(dts, keep) if k]
        dkp = dkp[keep]

        # cells = images with at least one instance or detection, in image order; detections of a cell in evaluation order
        img_ids = list(np.unique(p.imgIds))
        pos = {i: n for n, i in enumerate(img_ids)}
        g_img = np.array([pos[g["image_id"]] for g in gts], np.int64)
        d_img = np.array([pos[d["image_id"]] for d in dts], np.int64)
        d_score = np.array([d[self.score_key] for d in dts], np.float64)
        g_order = np.argsort(g_img, kind="mergesort")
        d_order = np.lexsort((-d_score, d_img)) if len(dts) else np.zeros(0, np.int64)  # stable: image, then descending score
        rank = np.zeros(len(dts), np.int64)
        if len(dts):
            sorted_img = d_img[d_order]
            start = np.r_[0, np.flatnonzero(np.diff(sorted_img)) + 1]
            rank = np.arange(len(dts)) - np.repeat(start, np.diff(np.r_[start, len(dts)]))
            d_order = d_order[rank < p.maxDets[-1]]  # at most maxDets per image (:548-550, :741)
        gts = [gts[i] for i in g_order]
        dts = [dts[i] for i in d_order]
        gkp, gbb, g_ignore, dkp = gkp[g_order], gbb[g_order], g_ignore[g_order], dkp[d_order]
        g_img, d_img, d_score = g_img[g_order], d_img[d_order], d_score[d_order]
        cells = np.unique(np.r_[g_img, d_img]).astype(np.int64)
        cell_gt_off = np.searchsorted(g_img, np.r_[cells, len(img_ids)], side="left").astype(np.int32)
        cell_dt_off = np.searchsorted(d_img, np.r_[cells, len(img_ids)], side="left").astype(np.int32)
        if len(cells):
            cell_gt_off[-1], cell_dt_off[-1] = len(gts), len(dts)
        else:
            cell_gt_off, cell_dt_off = np.zeros(1, np.int32), np.zeros(1, np.int32)
        Gc, Dc = np.diff(cell_gt_off).astype(np.int64), np.diff(cell_dt_off).astype(np.int64)
        cell_iou_off = np.r_[0, np.cumsum(L * Gc * Dc)].astype(np.int64)
        N_gt, N_dt = len(gts), len(dts)
        f64 = lambda rows, width: np.array(rows, np.float64).reshape(len(rows), *width)  # noqa: E731
        h = dict(
            gt_kpts=gkp, gt_bbox=gbb,
            gt_area_oks=f64([g.get("area", 0.0) for g in gts], ()),
            gt_area_rng=f64([g["area"] if ("area" in g and self.use_area) else g["bbox"][2] * g["bbox"][3] * 0.53 for g in gts], ()),
            gt_ignore=g_ignore.astype(np.uint8).reshape(N_gt, L),
            gt_iscrowd=np.array([int(g["iscrowd"]) for g in gts], np.uint8),
            dt_kpts=dkp, dt_bbox=f64([d["bbox"] for d in dts], (4,)),
            dt_area=f64([d["area"] for d in dts], ()), dt_score=d_score,
            order=np.argsort(-d_score, kind="mergesort").astype(np.int32),  # over the concatenation of the cells (:946-952)
            cell_gt_off=cell_gt_off, cell_dt_off=cell_dt_off, cell_iou_off=cell_iou_off[:-1].copy(),
            area_rng=np.array(p.areaRng, np.float64), iou_thrs=np.asarray(p.iouThrs, np.float64),
            rec_thrs=np.asarray(p.recThrs, np.float64), sigmas=self.sigmas, gt_vis=np.array(self.gt_visibilities, np.int32),
        )
        counts = np.stack([(h["gt_kpts"][:, :, 2] > 0).sum(1)] + [(h["gt_kpts"][:, :, 2] == v).sum(1) for v in self.gt_visibilities], 1) \
            if N_gt else np.zeros((0, L), np.int64)
        assert not bool((h["gt_ignore"].astype(bool) & (counts > 0)).any()), "k1 is negative but gt is not ignored"  # :654
        self._gts_flat, self._dts_flat, self._matched_pairs = gts, dts, None  # (flat evaluation order; matched_pairs reads them)
        self._meta = dict(n_cells=len(cells), N_gt=N_gt, N_dt=N_dt, L=L, K=K, iou_total=int(cell_iou_off[-1]),
                          max_g=int(Gc.max()) if len(Gc) else 0, gt_ids=[g.get("id") for g in gts], dt_ids=[d.get("id") for d in dts],
                          gt_img=[img_ids[i] for i in g_img], dt_img=[img_ids[i] for i in d_img])
        return h

    def evaluate(self):
        """Similarities + matching of every image at every level / area range / threshold (_cocoeval.py:424-503)."""
        if self.device.type != "cuda":
            raise RuntimeError("probpose_code_amd.evaluation.COCOeval runs on the GPU only (no CPU fallback)")
        assert self.padding >= 1.0, "Padding must be greater than or equal to 1.0"  # :560
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        p.maxDets = sorted(p.maxDets)
        h = self._prepare()
        m = self._meta
        dev = self.device
        d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in h.items()}
        L, A, T = m["L"], len(p.areaRng), len(p.iouThrs)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ious = torch.empty(max(m["iou_total"], 1), dtype=torch.float64, device=dev)
        thr = float("nan") if self.confidence_thr is None else float(self.confidence_thr)
        ptr = lambda t: t.data_ptr() if t.numel() else None  # noqa: E731
        if m["n_cells"] and m["iou_total"]:
            _lib.call("pp_exoks_cells", ptr(d["gt_kpts"]), ptr(d["gt_bbox"]), ptr(d["gt_area_oks"]), ptr(d["dt_kpts"]),
                      d["sigmas"].data_ptr(), ptr(d["gt_vis"]), d["cell_gt_off"].data_ptr(), d["cell_dt_off"].data_ptr(),
                      d["cell_iou_off"].data_ptr(), m["n_cells"], m["K"], L - 1, thr, float(self.padding), int(self.use_area),
                      int(not self.extended_oks), ious.data_ptr(), stream)
        o = dict(
            dt_match=torch.full((L, A, T, m["N_dt"]), -1, dtype=torch.int32, device=dev),
            dt_ignore=torch.zeros((L, A, T, m["N_dt"]), dtype=torch.uint8, device=dev),
            gt_match=torch.full((L, A, T, m["N_gt"]), -1, dtype=torch.int32, device=dev),
            gt_ignore=torch.zeros((L, A, m["N_gt"]), dtype=torch.uint8, device=dev),
            sim_sum=torch.zeros((L, A, max(m["n_cells"], 1)), dtype=torch.float64, device=dev),
            sim_cnt=torch.zeros((L, A, max(m["n_cells"], 1)), dtype=torch.int32, device=dev),
        )
        if m["n_cells"]:
            _lib.call("pp_exoks_match", ious.data_ptr(), d["cell_gt_off"].data_ptr(), d["cell_dt_off"].data_ptr(),
                      d["cell_iou_off"].data_ptr(), ptr(d["gt_ignore"]), ptr(d["gt_iscrowd"]), ptr(d["gt_area_rng"]), ptr(d["gt_bbox"]),
                      ptr(d["dt_area"]), ptr(d["dt_bbox"]), d["area_rng"].data_ptr(), d["iou_thrs"].data_ptr(), m["n_cells"],
                      m["max_g"], m["N_gt"], m["N_dt"], L, A, T, int(self.match_by_bbox), ptr(o["dt_match"]), ptr(o["dt_ignore"]),
                      ptr(o["gt_match"]), ptr(o["gt_ignore"]), o["sim_sum"].data_ptr(), o["sim_cnt"].data_ptr(), stream)
        self._d, self._o, self.ious = d, o, ious
        s, n = o["sim_sum"].cpu().numpy(), o["sim_cnt"].cpu().numpy()
        self.loc_similarity_mean = float(s.sum() / n.sum()) if n.sum() else float("nan")  # np.mean(loc_similarities), :1186
        self.n_loc_similarities = int(n.sum())

    def accumulate(self, p=None):
        """Precision / recall / score tables (_cocoeval.py:889-1009)."""
        if self._d is None:
            raise Exception("Please run evaluate() first")
        p = self.params if p is None else p
        m, d, o = self._meta, self._d, self._o
        L, A, T, R = m["L"], len(p.areaRng), len(p.iouThrs), len(p.recThrs)
        dev = self.device
        precision = torch.full((T, L, R, A), -1.0, dtype=torch.float64, device=dev)
        recall = torch.full((T, L, A), -1.0, dtype=torch.float64, device=dev)
        scores = torch.full((T, L, R, A), -1.0, dtype=torch.float64, device=dev)
        scratch = torch.empty((L * A * T, max((m["N_dt"] + 255) // 256, 1), 2), dtype=torch.int32, device=dev)
        ptr = lambda t: t.data_ptr() if t.numel() else None  # noqa: E731
        _lib.call("pp_exmap_accumulate", ptr(o["dt_match"]), ptr(o["dt_ignore"]), ptr(o["gt_ignore"]), ptr(d["order"]),
                  ptr(d["dt_score"]), d["rec_thrs"].data_ptr(), m["n_cells"], m["N_gt"], m["N_dt"], L, A, T, R, scratch.data_ptr(),
                  precision.data_ptr(), recall.data_ptr(), scores.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        self.eval = {
            "params": p, "counts": [T, L, R, 1, A, 1],
            "precision": precision.cpu().numpy().reshape(T, L, R, 1, A, 1),
            "recall": recall.cpu().numpy().reshape(T, L, 1, A, 1),
            "scores": scores.cpu().numpy().reshape(T, L, R, 1, A, 1),
        }

    def summarize(self):
        """The 11 + len(gt_visibilities) numbers CocoMetric reports (_cocoeval.py:1017-1059, :1136-1190)."""
        if not self.eval:
            raise Exception("Please run accumulate() first")
        p = self.params

        def mean_of(ap, iouThr=None, areaRng="all", visibility=None):
            a = p.areaRngLbl.index(areaRng)
            v = 0 if visibility is None else self.gt_visibilities.index(visibility) + 1
            s = self.eval["precision"] if ap else self.eval["recall"]
            if iouThr is not None:
                s = s[np.where(iouThr == p.iouThrs)[0]]
            s = s[:, v, :, :, a, 0] if ap else s[:, v, :, a, 0]
            s = s[s > -1]
            return -1 if len(s) == 0 else np.mean(s)

        names, stats = ["AP"], [mean_of(1)]
        for v in self.gt_visibilities:
            names.append("AP (v={:d})".format(v))
            stats.append(mean_of(1, visibility=v))
        for tag, ap in (("AP", 1), ("AR", 0)):
            if not ap:
                names.append("AR")
                stats.append(mean_of(0))
            names += [tag + " .5", tag + " .75", tag + " (M)", tag + " (L)"]
            stats += [mean_of(ap, iouThr=0.5), mean_of(ap, iouThr=0.75), mean_of(ap, areaRng="medium"), mean_of(ap, areaRng="large")]
        names.append("OKS")
        stats.append(self.loc_similarity_mean)
        self.stats, self.stats_names = np.array(stats, np.float64), names

    @property
    def matched_pairs(self):
        """``COCOeval.matched_pairs`` (_cocoeval.py:486-499 with the ``return_matching and match_by_bbox`` branch of
        ``evaluateImg``, :762-777): for every image, detections in score order (at most maxDets) against the instances
        in not-ignored-first order (level 0, all areas); the first instance whose box centre lies within an L1 distance of
        2 px of the detection's is its partner. List of ``(detection dict, instance dict, similarity)``, similarity =
        level-0 entry of the similarity matrix the GPU computed, ``nan`` for an instance ignored at level 0. Host
        bookkeeping over the device-computed matrix, derived on first use."""
        if self._d is None:
            raise Exception("Please run evaluate() first")
        if self._matched_pairs is not None:
            return self._matched_pairs
        m, h = self._meta, self._d
        ious = self.ious.cpu().numpy()
        g_off, d_off = h["cell_gt_off"].cpu().numpy(), h["cell_dt_off"].cpu().numpy()
        i_off = h["cell_iou_off"].cpu().numpy()
        g_ign0 = h["gt_ignore"].cpu().numpy().reshape(m["N_gt"], m["L"])[:, 0].astype(bool) if m["N_gt"] else np.zeros(0, bool)
        g_area = h["gt_area_rng"].cpu().numpy()
        gbb, dbb = h["gt_bbox"].cpu().numpy().reshape(-1, 4), h["dt_bbox"].cpu().numpy().reshape(-1, 4)
        pairs = []
        for c in range(m["n_cells"]):
            g0, g1, d0, d1 = int(g_off[c]), int(g_off[c + 1]), int(d_off[c]), int(d_off[c + 1])
            G, D = g1 - g0, d1 - d0
            if G == 0 or D == 0:
                continue
            ign = g_ign0[g0:g1] | (g_area[g0:g1] < 0) | (g_area[g0:g1] > 1e5 ** 2)  # (:729-735 with aRng = [0, 1e5 ** 2])
            gtind = np.argsort(ign.astype(np.int64), kind="mergesort")
            sim0 = ious[int(i_off[c]):int(i_off[c]) + D * G].reshape(D, G)  # level 0 block: (detection, instance)
            gc = gbb[g0:g1, :2] + gbb[g0:g1, 2:] / 2
            for di in range(D):
                dc = dbb[d0 + di, :2] + dbb[d0 + di, 2:] / 2
                for gi in gtind:
                    if np.abs(dc - gc[gi]).sum() < 2:
                        iou = float("nan") if g_ign0[g0 + gi] else float(sim0[di, gi])
                        pairs.append((self._dts_flat[d0 + di], self._gts_flat[g0 + gi], iou))
                        break
        self._matched_pairs = pairs
        return pairs

    def image_results(self):
        """Per-image outcome of evaluate() as host arrays keyed like the reference's evalImgs entries: dt_match / gt_match
        hold annotation ids (-1 = unmatched), shaped (L, A, T, N); plus dt_ids / gt_ids / dt_img / gt_img (flat order)."""
        m, o = self._meta, self._o
        gt_ids, dt_ids = np.array(m["gt_ids"] + [-1]), np.array(m["dt_ids"] + [-1])
        return dict(dt_match=gt_ids[o["dt_match"].cpu().numpy()], dt_ignore=o["dt_ignore"].cpu().numpy().astype(bool),
                    gt_match=dt_ids[o["gt_match"].cpu().numpy()], gt_ignore=o["gt_ignore"].cpu().numpy().astype(bool),
                    dt_ids=dt_ids[:-1], gt_ids=gt_ids[:-1], dt_img=np.array(m["dt_img"]), gt_img=np.array(m["gt_img"]))


# ---------------------------------------------------------------------------------------------------------------------
# The metric driver around the evaluator (CocoMetric, mmpose/evaluation/metrics/coco_metric.py): host bookkeeping
# ---------------------------------------------------------------------------------------------------------------------
def oks_iou(g, d, a_g, a_d, sigmas=None, vis_thr=None):
    """OKS of instance ``g`` (K*3,) to instances ``d`` (N, K*3) with the mean of both areas as scale
    (mmpose/evaluation/functional/nms.py:58-116). float32 result like the reference."""
    var = ((COCO_SIGMAS if sigmas is None else np.asarray(sigmas)) * 2) ** 2
    d = np.asarray(d).reshape(len(d), len(g))
    out = np.zeros(len(d), np.float32)
    for n in range(len(d)):
        e = ((d[n, 0::3] - g[0::3]) ** 2 + (d[n, 1::3] - g[1::3]) ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if vis_thr is not None:
            e = e[(g[2::3] > vis_thr) & (d[n, 2::3] > vis_thr)]
        out[n] = np.sum(np.exp(-e)) / len(e) if len(e) != 0 else 0.0
    return out


def oks_nms(kpts_db, thr, sigmas=None, vis_thr=None, score_per_joint=False):
    """Greedy OKS suppression inside one image (nms.py:119-170): highest score first, drop what overlaps it by more than
    ``thr``. Returns the indices kept."""
    if len(kpts_db) == 0:
        return []
    scores = np.array([k["score"].mean() if score_per_joint else k["score"] for k in kpts_db])
    kpts = np.array([np.asarray(k["keypoints"]).flatten() for k in kpts_db])
    areas = np.array([k["area"] for k in kpts_db])
    order = scores.argsort()[::-1]
    keep = []
    while len(order) > 0:
        i = order[0]
        keep.append(i)
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, vis_thr)
        order = order[np.where(ovr <= thr)[0] + 1]
    return np.array(keep)


def nms(dets, thr):
    """Greedy box suppression of the multi-person demo (mmpose/evaluation/functional/nms.py:16-55): ``dets`` (N, 5)
    [x1, y1, x2, y2, score]; keep the highest score, drop boxes that overlap it by more than ``thr`` (IoU with the
    reference's +1 pixel convention). Returns the indices kept."""
    dets = np.asarray(dets)
    if len(dets) == 0:
        return []
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while len(order) > 0:
        i, rest = order[0], order[1:]
        keep.append(i)
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        order = rest[inter / (areas[i] + areas[rest] - inter) <= thr]
    return keep


def soft_oks_nms(kpts_db, thr, max_dets=20, sigmas=None, vis_thr=None, score_per_joint=False):
    """Soft OKS suppression (nms.py:173-259): the highest score is kept, the scores of the rest decay by
    exp(-oks^2 / thr) (gaussian rescoring), re-sort, repeat until ``max_dets`` are kept. Returns the indices kept, in
    the order they were picked."""
    if len(kpts_db) == 0:
        return []
    scores = np.array([k["score"].mean() if score_per_joint else k["score"] for k in kpts_db])
    kpts = np.array([np.asarray(k["keypoints"]).flatten() for k in kpts_db])
    areas = np.array([k["area"] for k in kpts_db])
    order = scores.argsort()[::-1]
    scores = scores[order]
    keep = []
    while len(order) > 0 and len(keep) < max_dets:
        i = order[0]
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]], sigmas, vis_thr)
        order = order[1:]
        scores = scores[1:] * np.exp(-(ovr ** 2) / thr)
        tmp = scores.argsort()[::-1]
        order, scores = order[tmp], scores[tmp]
        keep.append(i)
    return np.array(keep, dtype=np.intp)


def instance_score(bbox_score, keypoint_scores, keypoint_probs, score_mode="bbox_keypoint", score_thresh_type="score",
                   keypoint_score_thr=0.2):
    """Detection score of one instance (coco_metric.py:549-572)."""
    keypoint_scores = np.asarray(keypoint_scores)
    if score_mode == "bbox":
        return bbox_score
    if score_mode == "keypoint":
        return np.mean(keypoint_scores)
    if score_mode == "bbox_rle":
        return float(bbox_score + np.mean(keypoint_scores) + np.max(keypoint_scores))
    gate = keypoint_scores if score_thresh_type == "score" else np.asarray(keypoint_probs)
    sel = gate > keypoint_score_thr
    mean_kpt = float(np.sum(keypoint_scores[sel])) / int(sel.sum()) if sel.any() else 0
    return bbox_score * mean_kpt


def best_threshold(gt_labels, dt_values):
    """Accuracy-maximising threshold among 21 evenly spaced ones (coco_metric.py:1267-1270, 1308-1319, the deterministic
    ``force_balance=False`` branch). ``gt_labels``: 1 / 0 / NaN (NaN = not evaluated). Returns (accuracy, threshold)."""
    gt_labels, dt_values = np.asarray(gt_labels, np.float32), np.asarray(dt_values, np.float32)
    mask = ~np.isnan(gt_labels)
    g, d = gt_labels[mask].astype(bool), dt_values[mask]
    thresholds = np.linspace(0, 1.00, 21, endpoint=True)
    acc = np.sum((d[:, None] > thresholds) == g[:, None], axis=0) / len(g)
    i = int(np.argmax(acc))
    return acc[i], thresholds[i]


class CocoMetric:
    """``process()`` / ``compute_metrics()`` of the reference's CocoMetric (coco_metric.py:236-358, 459-628, 671-750) for
    top-down keypoint results and a ground truth given as COCO-style annotation dicts (or a COCO-API object): group the
    predictions per image, drop duplicates, score the instances, optional OKS suppression, then one evaluator run per
    (extended, match_by_bbox, ignore_border_points) setting with the reference's name prefixes ("Ex_", "bbox_",
    "_NoBrd"). ``prob_thr`` (the presence-probability threshold Ex-OKS binarises with): the reference takes the
    accuracy-maximising threshold of its classification analysis (coco_metric.py:981-1003); here it is
    ``best_threshold`` over the instances that have a ground truth of the same (image, id), or the value passed in.
    Not provided: the json dumps, converters, and the analysis printouts (calibration, OKS-to-IoU, vector fields)."""

    def __init__(self, gt_annotations, use_area=True, iou_type="keypoints", score_mode="bbox_keypoint", score_thresh_type="score",
                 keypoint_score_thr=0.2, nms_mode="oks_nms", nms_thr=0.9, prefix=None, extended=(False,), match_by_bbox=(False,),
                 ignore_border_points=(False,), ignore_stats=(), padding=1.25, prob_thr=None, sigmas=None, device="cuda"):
        if score_mode not in ("bbox", "bbox_keypoint", "bbox_rle", "keypoint"):
            raise ValueError(f"`score_mode` should be one of 'bbox', 'bbox_keypoint', 'bbox_rle', but got {score_mode}")
        if score_thresh_type not in ("score", "prob"):
            raise ValueError("'score_thresh_type' should be one of 'score' or 'prob'")
        if nms_mode not in ("oks_nms", "soft_oks_nms", "none"):
            raise ValueError("`nms_mode` should be one of 'oks_nms', 'soft_oks_nms', " f"'none', but got {nms_mode}")
        extended, match_by_bbox, ignore_border_points = list(extended), list(match_by_bbox), list(ignore_border_points)
        n = max(len(extended), len(match_by_bbox))
        if len(extended) == 1 and n > 1:
            extended = extended * n
        if len(match_by_bbox) == 1 and n > 1:
            match_by_bbox = match_by_bbox * n
        assert len(extended) == len(match_by_bbox), "The length of `extended` and `match_by_bbox` should be the same."
        if len(ignore_border_points) == 1 and n > 1:
            ignore_border_points = ignore_border_points * n
        self.gt, self.use_area, self.iou_type = gt_annotations, use_area, iou_type
        self.score_mode, self.score_thresh_type, self.keypoint_score_thr = score_mode, score_thresh_type, keypoint_score_thr
        self.nms_mode, self.nms_thr, self.prefix = nms_mode, nms_thr, prefix
        self.extended, self.match_by_bbox, self.ignore_border_points = extended, match_by_bbox, ignore_border_points
        self.ignore_stats, self.padding, self.prob_thr = list(ignore_stats), padding, prob_thr
        self.sigmas = COCO_SIGMAS if sigmas is None else np.asarray(sigmas, np.float64)
        self.device = device
        self.has_probability = True
        self.results = []

    def process(self, data_batch, data_samples):
        """One batch of ``PoseDataSample``-like dicts (``to_dict()``) -> ``self.results`` (coco_metric.py:236-358)."""
        for s in data_samples:
            if "pred_instances" not in s:
                raise ValueError(f"`pred_instances` are required to process the predictions results in {self.__class__.__name__}. ")
            pi = s["pred_instances"]
            kp = np.asarray(pi["keypoints"])
            sc = np.asarray(pi["keypoint_scores"])
            assert sc.shape == kp.shape[:2]
            if "keypoints_probs" not in pi:
                self.has_probability = False
            pred = dict(id=s["id"], img_id=s["img_id"], category_id=s.get("category_id", 1), keypoints=kp, keypoint_scores=sc,
                        keypoints_visible=np.asarray(pi.get("keypoints_visible", sc)), keypoint_probs=np.asarray(pi.get("keypoints_probs", sc)))
            if "bboxes" in pi:
                b = np.asarray(pi["bboxes"], np.float64).reshape(-1, 4)
                pred["bbox"] = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)  # bbox_xyxy2xywh
            gi = s.get("gt_instances", {})
            if "bbox_scores" in pi:
                pred["bbox_scores"] = np.asarray(pi["bbox_scores"])
            elif "bbox_scores" not in gi or len(gi["bbox_scores"]) != len(kp):
                pred["bbox_scores"] = np.ones(len(kp))
            else:
                pred["bbox_scores"] = np.asarray(gi["bbox_scores"])
            if "bbox_scales" in gi:
                pred["areas"] = np.prod(np.asarray(gi["bbox_scales"]), axis=1)
            self.results.append(pred)

    def _instances(self):
        per_img = {}
        for pred in self.results:  # coco_metric.py:495-529
            for i, kp in enumerate(pred["keypoints"]):
                inst = dict(id=pred["id"], img_id=pred["img_id"], category_id=pred["category_id"], keypoints=kp,
                            keypoint_scores=pred["keypoint_scores"][i], bbox_score=pred["bbox_scores"][i],
                            keypoints_visible=pred["keypoints_visible"][i], keypoint_probs=pred["keypoint_probs"][i])
                if "bbox" in pred:
                    inst["bbox"] = pred["bbox"][i]
                if "areas" in pred:
                    inst["area"] = pred["areas"][i]
                else:
                    inst["area"] = (np.max(kp[:, 0]) - np.min(kp[:, 0])) * (np.max(kp[:, 1]) - np.min(kp[:, 1]))
                per_img.setdefault(pred["img_id"], []).append(inst)
        for img_id, persons in per_img.items():  # _sort_and_unique_bboxes (:1321-1348)
            persons = sorted(persons, key=lambda x: x["id"])
            per_img[img_id] = [p for n, p in enumerate(persons) if n == 0 or p["id"] != persons[n - 1]["id"]]
        valid = {}
        for img_id, persons in per_img.items():  # :541-583
            for p in persons:
                p["keypoints"] = np.concatenate([p["keypoints"], np.asarray(p["keypoint_probs"])[:, None]], axis=-1)
                p["score"] = instance_score(p["bbox_score"], p["keypoint_scores"], p["keypoint_probs"], self.score_mode,
                                            self.score_thresh_type, self.keypoint_score_thr)
            if self.nms_mode == "none":
                valid[img_id] = persons
            else:
                nms = oks_nms if self.nms_mode == "oks_nms" else soft_oks_nms  # coco_metric.py:576
                valid[img_id] = [persons[k] for k in nms(persons, self.nms_thr, sigmas=self.sigmas)]
        return valid

    def _gt_list(self):
        if hasattr(self.gt, "loadAnns"):
            return self.gt.loadAnns(self.gt.getAnnIds(imgIds=self.gt.getImgIds()))
        return list(self.gt)

    def compute_metrics(self):
        valid = self._instances()
        gts = self._gt_list()
        dts = []
        # results2json (:630-669) + COCO.loadRes of the un-vendored xtcocotools / pycocotools, whose keypoint branch sets id,
        # area and bbox of every result from the extent of its keypoints
        for img_id, persons in valid.items():
            for p in persons:
                kp = np.asarray(p["keypoints"], np.float64)
                x, y = kp[:, 0], kp[:, 1]
                x0, x1, y0, y1 = float(x.min()), float(x.max()), float(y.min()), float(y.max())
                d = dict(image_id=p["img_id"], category_id=p["category_id"], keypoints=kp.flatten().tolist(), score=float(p["score"]),
                         id=len(dts) + 1, area=(x1 - x0) * (y1 - y0), bbox=[x0, y0, x1 - x0, y1 - y0])
                dts.append(d)
        out = {}
        if self.prob_thr is None:  # the probability threshold of the classification analysis (:949-1003)
            by_key = {(g["image_id"], g["id"]): g for g in gts if not np.allclose(np.array(g["keypoints"]), 0)}
            labels, probs, scores = [], [], []
            for persons in valid.values():
                for p in persons:
                    g = by_key.get((p["img_id"], p["id"]))
                    if g is None:
                        continue
                    v = np.array(g["keypoints"], np.float64)[2::3]
                    labels.append(np.where(v == 0, np.nan, np.where(v == 3, 0.0, 1.0)))
                    probs.append(p["keypoint_probs"])
                    scores.append(p["keypoint_scores"])
            labels = np.array(labels).flatten()
            if len(np.unique(labels[~np.isnan(labels)])) > 1:
                acc, thr = best_threshold(labels, np.array(probs if self.has_probability else scores).flatten())
                out["prob_acc" if self.has_probability else "score_acc"] = float(acc)
                out["prob_thr" if self.has_probability else "score_thr"] = float(thr)
                self.prob_thr = float(thr)
            else:
                self.prob_thr = -1  # the reference's initial value (:186): every probability counts as "present"
        for ext, mbb, nobrd in zip(self.extended, self.match_by_bbox, self.ignore_border_points):  # :692-748
            prefix = ("Ex_" if ext else "") + ("bbox_" if mbb else "")
            suffix = "_NoBrd" if nobrd else ""
            e = COCOeval(gts, dts, self.iou_type, sigmas=self.sigmas, use_area=self.use_area, extended_oks=ext, match_by_bbox=mbb,
                         confidence_thr=self.prob_thr, padding=self.padding, ignore_near_bbox=nobrd, device=self.device)
            e.evaluate()
            e.accumulate()
            e.summarize()
            for k, v in zip(e.stats_names, e.stats):
                if k not in self.ignore_stats:
                    out[f"{prefix}{k}{suffix}"] = float(v)
        if self.prefix:
            out = {f"{self.prefix}/{k}": v for k, v in out.items()}
        return out
