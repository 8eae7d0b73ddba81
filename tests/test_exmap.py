"""Ex-mAP evaluator: oracle vs the reference's own outputs (CPU), the HIP evaluator vs both (GPU).

Index / count work: every comparison is exact (==) except the mean localisation similarity, a float64 sum whose order of
summation differs (1e-12)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = np.load(os.path.join(HERE, "golden", "exmap_cases.npz"))
N = int(CASES["n_cases"])
K = 17
SIGMAS = np.array([0.26, 0.25, 0.25, 0.35, 0.35, 0.79, 0.79, 0.72, 0.72, 0.62, 0.62, 1.07, 1.07, 0.87, 0.87, 0.89, 0.89]) / 10.0
A, T = 3, 10


def _case(n):
    t = f"s{n}"
    ext, mbb, thr, padding, use_area, near = CASES[t + "/settings"]
    ptc = CASES[t + "/gt_ptc"] if t + "/gt_ptc" in CASES else None
    gts = []
    for i, (ids, kp, box) in enumerate(zip(CASES[t + "/gt_ids"], CASES[t + "/gt_kpts"], CASES[t + "/gt_box"])):
        g = dict(id=int(ids[0]), image_id=int(ids[1]), category_id=1, iscrowd=int(ids[2]), keypoints=kp.tolist(),
                 bbox=box[:4].tolist(), area=float(box[4]))
        if ptc is not None:
            g["pad_to_contain"] = ptc[i].tolist()
        gts.append(g)
    dts = [dict(id=int(ids[0]), image_id=int(ids[1]), category_id=1, keypoints=kp.tolist(), bbox=box[:4].tolist(),
                area=float(box[4]), score=float(box[5]))
           for ids, kp, box in zip(CASES[t + "/dt_ids"], CASES[t + "/dt_kpts"], CASES[t + "/dt_box"])]
    kw = dict(use_area=bool(use_area), extended_oks=bool(ext), match_by_bbox=bool(mbb),
              confidence_thr=None if np.isnan(thr) else float(thr), padding=float(padding), ignore_near_bbox=bool(near))
    return gts, dts, CASES[t + "/img_ids"].tolist(), kw


def _check_tables(n, precision, recall, scores, stats, names, gt_vis, n_sims):
    t = f"s{n}"
    assert list(gt_vis) == CASES[t + "/gt_visibilities"].tolist()
    assert np.array_equal(precision, CASES[t + "/precision"]), (n, "precision")
    assert np.array_equal(recall, CASES[t + "/recall"]), (n, "recall")
    assert np.array_equal(scores, CASES[t + "/scores"]), (n, "scores")
    assert list(names) == CASES[t + "/stats_names"].tolist()
    assert n_sims == int(CASES[t + "/n_loc_similarities"])
    ref = CASES[t + "/stats"]
    assert np.array_equal(stats[:-1], ref[:-1]), (n, stats, ref)
    assert abs(stats[-1] - ref[-1]) <= 1e-12


@pytest.mark.parametrize("n", range(N))
def test_oracle_matches_reference_evaluator(n):
    from oracle import exmap_ref

    gts, dts, img_ids, kw = _case(n)
    r = exmap_ref.evaluate(gts, dts, SIGMAS, img_ids=img_ids, **kw)
    _check_tables(n, r["precision"], r["recall"], r["scores"], r["stats"], r["stats_names"], r["gt_visibilities"],
                  len(r["loc_similarities"]))
    rows_dt, rows_gt = [], []
    for idx, e in enumerate(r["eval_imgs"]):
        if e is None:
            continue
        for di, d_id in enumerate(e["dtIds"]):
            rows_dt.append([idx, d_id] + e["dtMatches"][:, di].tolist() + np.asarray(e["dtIgnore"])[:, di].astype(int).tolist())
        for gi, g_id in enumerate(e["gtIds"]):
            rows_gt.append([idx, g_id, int(e["gtIgnore"][gi])] + e["gtMatches"][:, gi].tolist())
    t = f"s{n}"
    assert np.array_equal(np.array(rows_dt).reshape(-1, 22), CASES[t + "/img_dt_rows"])
    assert np.array_equal(np.array(rows_gt).reshape(-1, 13), CASES[t + "/img_gt_rows"])
    assert np.array_equal(np.array([e is None for e in r["eval_imgs"]]), CASES[t + "/img_none"])


def test_host_interface_needs_the_gpu():
    from probpose_code_amd.evaluation import COCOeval

    gts, dts, _, kw = _case(4)
    e = COCOeval(gts, dts, "keypoints", device="cpu", **kw)
    assert e.params.maxDets == [20] and len(e.params.iouThrs) == 10 and len(e.params.recThrs) == 101
    with pytest.raises(RuntimeError):
        e.evaluate()
    with pytest.raises(Exception):
        COCOeval(gts, dts, "bbox")


def _run(n):
    from probpose_code_amd.evaluation import COCOeval

    gts, dts, img_ids, kw = _case(n)
    e = COCOeval(gts, dts, "keypoints", sigmas=SIGMAS, **kw)
    e.params.imgIds = img_ids
    e.evaluate()
    e.accumulate()
    e.summarize()
    return e, img_ids


@pytest.mark.gpu
@pytest.mark.parametrize("n", range(N))
def test_hip_evaluator_matches_reference_tables(lib_built, n):
    e, _ = _run(n)
    _check_tables(n, e.eval["precision"], e.eval["recall"], e.eval["scores"], e.stats, e.stats_names, e.gt_visibilities,
                  e.n_loc_similarities)


@pytest.mark.gpu
@pytest.mark.parametrize("n", range(N))
def test_hip_matching_matches_reference_per_image(lib_built, n):
    e, img_ids = _run(n)
    r = e.image_results()
    t = f"s{n}"
    n_img = len(img_ids)
    img_pos = {i: k for k, i in enumerate(img_ids)}
    dt_at = {int(i): k for k, i in enumerate(r["dt_ids"])}
    gt_at = {int(i): k for k, i in enumerate(r["gt_ids"])}
    rows = CASES[t + "/img_dt_rows"]
    assert len(rows) == r["dt_match"].shape[0] * A * len(r["dt_ids"])  # every kept detection appears in every (level, area) list
    for row in rows:
        idx, d_id = int(row[0]), int(row[1])
        la, img = divmod(idx, n_img)
        lvl, a = divmod(la, A)
        j = dt_at[d_id]
        assert img_pos[int(r["dt_img"][j])] == img
        assert np.array_equal(r["dt_match"][lvl, a, :, j], row[2:12]), (idx, d_id)
        assert np.array_equal(r["dt_ignore"][lvl, a, :, j].astype(int), row[12:22]), (idx, d_id)
    for row in CASES[t + "/img_gt_rows"]:
        idx, g_id = int(row[0]), int(row[1])
        la, img = divmod(idx, n_img)
        lvl, a = divmod(la, A)
        j = gt_at[g_id]
        assert int(r["gt_ignore"][lvl, a, j]) == int(row[2])
        assert np.array_equal(r["gt_match"][lvl, a, :, j], row[3:13]), (idx, g_id)


@pytest.mark.gpu
@pytest.mark.parametrize("n", range(N))
def test_matched_pairs_equal_the_reference(lib_built, n):
    """COCOeval.matched_pairs (_cocoeval.py:486-499): same (detection id, instance id) pairs in the same order as the
    reference's evaluator produced for this dataset, similarities to 1e-9, nan where the instance is ignored at level 0."""
    e, _ = _run(n)
    want = CASES[f"s{n}/matched_pairs"]
    got = np.array([[d["id"], g["id"], iou] for d, g, iou in e.matched_pairs], np.float64).reshape(-1, 3)
    assert got.shape == want.shape and len(want) > 0
    assert np.array_equal(got[:, :2], want[:, :2])
    assert np.array_equal(np.isnan(got[:, 2]), np.isnan(want[:, 2]))
    ok = ~np.isnan(want[:, 2])
    assert np.abs(got[ok, 2] - want[ok, 2]).max() <= 1e-9
    assert e.matched_pairs is e.matched_pairs  # derived once


@pytest.mark.gpu
def test_hip_evaluator_matches_oracle_on_a_larger_dataset(lib_built):
    """~2000 detections (8 accumulate chunks), crowded images, ties; the oracle is the checker."""
    from oracle import exmap_ref
    from probpose_code_amd.evaluation import COCOeval

    rng = np.random.default_rng(7)
    gts, dts = [], []
    for img in range(220):
        G = int(rng.integers(0, 7))
        here = []
        for _ in range(G):
            w, h = rng.uniform(30, 220), rng.uniform(40, 320)
            x0, y0 = rng.uniform(0, 640 - w), rng.uniform(0, 480 - h)
            kp = np.zeros((K, 3))
            kp[:, 0], kp[:, 1] = rng.uniform(x0, x0 + w, K), rng.uniform(y0, y0 + h, K)
            vis = rng.choice([0, 1, 2, 3], K, p=[0.2, 0.2, 0.45, 0.15])
            kp[:, 2] = vis
            kp[vis == 0, :2] = 0
            g = dict(id=len(gts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), bbox=[x0, y0, w, h],
                     area=float(w * h * 0.5), iscrowd=int(rng.random() < 0.1))
            gts.append(g)
            here.append(g)
        for _ in range(int(rng.integers(0, 24))):
            if here and rng.random() < 0.85:
                g = here[rng.integers(0, len(here))]
                kp = np.array(g["keypoints"]).reshape(K, 3).copy()
                kp[:, :2] += rng.normal(0, rng.choice([0.005, 0.02, 0.06]) * np.sqrt(g["bbox"][2] * g["bbox"][3]), (K, 2))
                kp[:, 2] = np.where(kp[:, 2] == 3, rng.beta(1.2, 4, K), rng.beta(5, 1.2, K))
                bbox = list(g["bbox"])
            else:
                kp = np.stack([rng.uniform(0, 640, K), rng.uniform(0, 480, K), rng.uniform(0, 1, K)], 1)
                bbox = [float(kp[:, 0].min()), float(kp[:, 1].min()), float(np.ptp(kp[:, 0])), float(np.ptp(kp[:, 1]))]
            dts.append(dict(id=len(dts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(),
                            score=float(np.round(rng.uniform(0.05, 1.0), 2)), bbox=bbox, area=float(bbox[2] * bbox[3])))
    img_ids = list(range(225))
    for mbb in (False, True):
        ref = exmap_ref.evaluate(gts, dts, SIGMAS, img_ids=img_ids, match_by_bbox=mbb)
        e = COCOeval(gts, dts, "keypoints", sigmas=SIGMAS, extended_oks=True, match_by_bbox=mbb)
        e.params.imgIds = img_ids
        e.evaluate()
        e.accumulate()
        e.summarize()
        assert e._meta["N_dt"] > 1800
        assert np.array_equal(e.eval["precision"], ref["precision"])
        assert np.array_equal(e.eval["recall"], ref["recall"])
        assert np.array_equal(e.eval["scores"], ref["scores"])
        assert np.array_equal(e.stats[:-1], ref["stats"][:-1]) and abs(e.stats[-1] - ref["stats"][-1]) <= 1e-12
        assert e.stats_names == ref["stats_names"]


@pytest.mark.gpu
def test_hip_evaluator_empty_and_degenerate_inputs(lib_built):
    from probpose_code_amd.evaluation import COCOeval

    gts, dts, img_ids, kw = _case(0)
    e = COCOeval(gts, [], "keypoints", sigmas=SIGMAS, **kw)  # no detections: recall 0, precision 0 where instances count
    e.evaluate(); e.accumulate(); e.summarize()
    assert e.stats[0] == 0.0 and np.isnan(e.stats[-1])
    e = COCOeval([], dts, "keypoints", sigmas=SIGMAS, **kw)  # no ground truth: nothing is evaluated
    e.params.imgIds = img_ids
    e.evaluate(); e.accumulate(); e.summarize()
    assert e.stats[0] == -1 and e.gt_visibilities == []
    with pytest.raises(AssertionError):
        COCOeval(gts, dts, "keypoints", sigmas=SIGMAS, extended_oks=True, padding=0.9).evaluate()


@pytest.mark.gpu
def test_differential_fuzz_against_the_oracle(lib_built):
    """tests/fuzz_exmap.py for a few seconds: random datasets (empty images, crowds, score ties, zero-area boxes, all-invisible annotations) through
    every switch combination, the HIP evaluator's tables equal to the oracle's."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_exmap.py"), "12"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "EXMAP FUZZ OK" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
