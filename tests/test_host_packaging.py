"""CPU: the host side of the drop-in call - `ProbMapHead.pack_records` (probmap_head.py:779-804 from the batch's one host
record) and `TopdownPoseEstimator.add_pred_to_datasample` (topdown.py:128-194) with the image-space map evaluated once per
batch - against the per-sample expressions of the reference, bit for bit. No engine: the head only needs the heatmap size."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")


class _Geometry:
    Hh, Wh, K = 64, 48, 17


@pytest.fixture()
def model():
    from probpose_code_amd import Config, build_pose_estimator

    m = dict(Config.fromfile(CFG).model)
    m.pop("train_cfg", None)
    model = build_pose_estimator(m)
    model._engine = _Geometry()
    return model


def _batch(B, rng, scale_dtype=np.float32):
    from probpose_code_amd import apis

    crops = torch.zeros(B, 3, 256, 192, dtype=torch.uint8)
    c = rng.uniform(50, 500, (B, 2)).astype(np.float32)
    s = rng.uniform(100, 700, (B, 2)).astype(np.float32)
    batch = apis.pack_crops(crops, c, s, apis.coco_dataset_meta())
    if scale_dtype != np.float32:
        for ds in batch["data_samples"]:
            ds.set_metainfo(dict(input_scale=np.asarray(ds.metainfo["input_scale"], scale_dtype)))
    return batch, c, s


def test_pack_records_fields_follow_the_reference_packaging(model):
    rng = np.random.default_rng(0)
    B, K = 5, 17
    kp = rng.uniform(0, 255, (B, K, 2))
    f32 = rng.random((B, K, 5)).astype(np.float32)  # conf, prob, vis, oks, raw error
    rec = np.concatenate([kp, f32.astype(np.float64)], -1)
    preds = model.head.pack_records(rec.copy(), model.test_cfg)
    assert len(preds) == B
    diag = np.sqrt(64**2 + 48**2)
    for b, p in enumerate(preds):
        assert p.keypoints.shape == (1, K, 2) and p.keypoints.dtype == np.float64 and np.array_equal(p.keypoints[0], kp[b])
        assert np.array_equal(p.keypoints_conf, f32[b, :, 0][None]) and p.keypoints_conf.dtype == np.float32
        assert np.array_equal(p.keypoints_probs, f32[b, :, 1][None])
        assert np.array_equal(p.keypoints_visible, f32[b, :, 2][None])
        assert np.array_equal(p.keypoints_oks, f32[b, :, 3][None])
        want_err = f32[b, :, 4].reshape(1, K) / diag  # probmap_head.py:786-787, numpy's own promotion
        assert np.array_equal(p.keypoints_error, want_err) and p.keypoints_error.dtype == want_err.dtype
        assert np.array_equal(p.keypoint_scores, f32[b, :, 3][None]), "freeze_oks=False: keypoint_scores <- oks (:797-798)"


@pytest.mark.parametrize("scale_dtype", [np.float32, np.float64])
def test_batched_image_space_map_equals_per_sample_expression(model, scale_dtype):
    rng = np.random.default_rng(1)
    B, K = 7, 17
    rec = np.concatenate([rng.uniform(0, 255, (B, K, 2)), rng.random((B, K, 5)).astype(np.float32).astype(np.float64)], -1)
    batch, c, s = _batch(B, rng, scale_dtype)
    preds = model.head.pack_records(rec.copy(), model.test_cfg)
    out = model.add_pred_to_datasample(preds, None, batch["data_samples"])
    for b, ds in enumerate(out):
        m = ds.metainfo
        want = rec[b:b + 1, :, :2] / m["input_size"] * m["input_scale"] + m["input_center"] - 0.5 * m["input_scale"]  # topdown.py:165-167
        assert np.array_equal(ds.pred_instances.keypoints, want) and ds.pred_instances.keypoints.dtype == np.float64
        assert np.array_equal(ds.pred_instances.bboxes, ds.gt_instances.bboxes)


def test_mixed_metainfo_dtypes_fall_back_to_the_per_sample_map(model):
    rng = np.random.default_rng(2)
    B, K = 4, 17
    rec = np.concatenate([rng.uniform(0, 255, (B, K, 2)), rng.random((B, K, 5)).astype(np.float32).astype(np.float64)], -1)
    batch, _, _ = _batch(B, rng)
    ds1 = batch["data_samples"][1]
    ds1.set_metainfo(dict(input_center=np.asarray(ds1.metainfo["input_center"], np.float64)))  # one sample differs in dtype
    preds = model.head.pack_records(rec.copy(), model.test_cfg)
    assert not model._map_batch_to_image_space(preds, batch["data_samples"])
    assert np.array_equal(preds[0].keypoints[0], rec[0, :, :2]), "a refused batch map must leave the keypoints untouched"
    out = model.add_pred_to_datasample(preds, None, batch["data_samples"])
    for b, ds in enumerate(out):
        m = ds.metainfo
        want = rec[b:b + 1, :, :2] / m["input_size"] * m["input_scale"] + m["input_center"] - 0.5 * m["input_scale"]
        assert np.array_equal(ds.pred_instances.keypoints, want)


def test_batched_and_per_sample_paths_agree_bit_for_bit_and_the_batch_is_mapped_once(model):
    """The two code paths of `add_pred_to_datasample` (one map per batch / the reference's per-sample expression) on the same inputs -
    float32 metainfo, float64 metainfo - give identical bits; and the batch array is marked as mapped: a second call on the
    same preds behaves like the reference's in-place assignment (it maps the given keypoints again, per sample), never a silent second
    batch map through the shared buffer."""
    rng = np.random.default_rng(3)
    B, K = 6, 17
    rec = np.concatenate([rng.uniform(0, 255, (B, K, 2)), rng.random((B, K, 5)).astype(np.float32).astype(np.float64)], -1)
    for variant in ("f32", "f64"):  # (input_size is a tuple in both, as PackPoseInputs hands it; center / scale are arrays as in the reference)
        batch, _, _ = _batch(B, np.random.default_rng(4), np.float64 if variant == "f64" else np.float32)
        fast = model.head.pack_records(rec.copy(), model.test_cfg)
        assert fast.keypoints_batch is not None
        out_fast = model.add_pred_to_datasample(fast, None, batch["data_samples"])
        assert fast.keypoints_batch is None, "the batch map must mark the shared array as mapped"
        kp_fast = np.stack([ds.pred_instances.keypoints for ds in out_fast]).copy()
        batch2, _, _ = _batch(B, np.random.default_rng(4), np.float64 if variant == "f64" else np.float32)
        slow = model.head.pack_records(rec.copy(), model.test_cfg)
        slow.keypoints_batch = None  # force the per-sample path
        out_slow = model.add_pred_to_datasample(slow, None, batch2["data_samples"])
        kp_slow = np.stack([ds.pred_instances.keypoints for ds in out_slow])
        assert kp_fast.dtype == kp_slow.dtype and np.array_equal(kp_fast, kp_slow), variant
        # second call on the already mapped preds: per sample, like the reference (topdown.py:165-167 assigns in place)
        again = model.add_pred_to_datasample(fast, None, batch["data_samples"])
        for b, ds in enumerate(again):
            m = ds.metainfo
            want = kp_fast[b] / np.asarray(m["input_size"]) * np.asarray(m["input_scale"]) + np.asarray(m["input_center"]) - 0.5 * np.asarray(m["input_scale"])
            assert np.array_equal(ds.pred_instances.keypoints, want)


def test_flip_modes_outside_the_path_are_rejected_with_the_reason(model):
    model.test_cfg = dict(flip_test=True, flip_mode="heatmap", shift_heatmap=True)
    assert model._check_flip_cfg() is True and model._shift_heatmap is True  # (tta.py:64-66: built into the fused flip merge)
    model.test_cfg = dict(flip_test=True, flip_mode="udp_combined")
    with pytest.raises(NotImplementedError, match="flip_mode='udp_combined'"):
        model._check_flip_cfg()
    model.test_cfg = dict(flip_test=False, shift_heatmap=True)
    assert model._check_flip_cfg() is False and model._shift_heatmap is False
