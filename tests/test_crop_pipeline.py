"""Crop pipeline in front of the hot path: box arithmetic against the reference's own outputs
(CPU), the HIP warp against the oracle restatement of cv2.warpAffine (GPU, bit-exact; the oracle itself is unpinned - no
cv2 here), inference_topdown end to end."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "warp_boxes.npz"))


def test_box_arithmetic_matches_reference_outputs():
    from probpose_code_amd import transforms as T

    assert np.array_equal(T.bbox_xywh2xyxy(G["boxes_xywh"]), G["xywh2xyxy"])
    for pad in (1.0, 1.25):
        c, s = T.bbox_xyxy2cs(G["boxes_xyxy"], padding=pad)
        assert np.array_equal(c, G[f"center_p{pad}"]) and np.array_equal(s, G[f"scale_p{pad}"])
    c, s = T.bbox_xyxy2cs(G["boxes_xyxy"], padding=1.25)
    fixed = T.fix_aspect_ratio(s, aspect_ratio=192 / 256)
    assert np.allclose(fixed, G["fixed_scale"], rtol=0, atol=0)
    for i in range(len(c)):
        m = T.get_udp_warp_matrix(c[i], G["fixed_scale"][i], float(G["rot"][i]), (192, 256))
        assert m.dtype == np.float32 and np.array_equal(m, G["udp_mats"][i]), i


def test_udp_warp_maps_box_corners_onto_the_input_grid():
    """UDP: the padded, aspect-fixed box maps onto [0, w-1] x [0, h-1] exactly."""
    from probpose_code_amd import transforms as T

    c, s, mats = T.topdown_affine_params(G["boxes_xyxy"][:5], (192, 256))
    for i in range(5):
        tl = mats[i] @ np.array([c[i, 0] - s[i, 0] / 2, c[i, 1] - s[i, 1] / 2, 1.0])
        br = mats[i] @ np.array([c[i, 0] + s[i, 0] / 2, c[i, 1] + s[i, 1] / 2, 1.0])
        assert np.allclose(tl, [0, 0], atol=2e-3) and np.allclose(br, [191, 255], atol=2e-3)
        assert np.allclose(T.invert_affine(mats[i]) @ np.append(mats[i] @ np.array([10.0, 20.0, 1.0]), 1.0), [10, 20], atol=1e-3)


def test_oracle_warp_identity_and_shift():
    """Known answers for the restated cv2 arithmetic: identity copies, an integer shift moves and zero-fills, a half-pixel
    shift averages neighbours with round-half-up."""
    from oracle import warp_ref

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    assert np.array_equal(warp_ref.warp_affine_u8(img, eye, (30, 20)), img)
    sh = warp_ref.warp_affine_u8(img, np.array([[1, 0, 3], [0, 1, 2]], np.float32), (30, 20))
    assert np.array_equal(sh[2:, 3:], img[:-2, :-3]) and not sh[:2].any() and not sh[:, :3].any()
    half = warp_ref.warp_affine_u8(img, np.array([[1, 0, 0.5], [0, 1, 0]], np.float32), (30, 20))
    exp = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + 16384) >> 15
    assert np.array_equal(half[:, 1:], exp.astype(np.uint8))


def test_oracle_warp_tracks_scipy_bilinear():
    """cv2 is absent, so the fixed-point restatement cannot be pinned to it. Independent cross-check: scipy's float
    bilinear resampling (`ndimage.map_coordinates(order=1)`) at the same source coordinates of a rotated + scaled UDP-like
    warp. cv2 quantises the interpolation weights to 1/32 px (INTER_BITS = 5), so the two may differ by the image gradient
    x 1/64 px plus rounding - bounded here by 2 grey levels + |gradient| / 32 on a smooth image, inside the valid region."""
    from scipy import ndimage

    from oracle import warp_ref

    yy, xx = np.mgrid[0:120, 0:160].astype(np.float64)
    img = (127 + 80 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 30 * np.sin((xx + yy) / 23.0)).round().astype(np.uint8)[..., None]
    th, sc = 0.3, 1.37
    m = np.array([[sc * np.cos(th), -sc * np.sin(th), 10.2], [sc * np.sin(th), sc * np.cos(th), -3.7]], np.float32)  # src -> dst
    out = warp_ref.warp_affine_u8(img, m, (96, 128))[..., 0].astype(np.float64)
    inv = warp_ref.invert_affine(m)
    oy, ox = np.mgrid[0:128, 0:96].astype(np.float64)
    sx, sy = inv[0, 0] * ox + inv[0, 1] * oy + inv[0, 2], inv[1, 0] * ox + inv[1, 1] * oy + inv[1, 2]
    ref = ndimage.map_coordinates(img[..., 0].astype(np.float64), [sy, sx], order=1, mode="constant", cval=0.0)
    valid = (sx > 1) & (sx < 158) & (sy > 1) & (sy < 118)
    gy, gx = np.gradient(img[..., 0].astype(np.float64))
    g = ndimage.map_coordinates(np.hypot(gx, gy), [sy, sx], order=1, mode="nearest")
    assert valid.mean() > 0.5
    assert (np.abs(out - ref)[valid] <= 2.0 + g[valid] / 32.0 * 2.0).all(), float(np.abs(out - ref)[valid].max())
    assert np.abs(out - ref)[valid].mean() < 0.5


@pytest.mark.gpu
def test_hip_warp_bit_exact_vs_oracle():
    import torch

    from oracle import warp_ref
    from probpose_code_amd import transforms as T

    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[0, 0, 640, 480], [100.3, 50.7, 220.9, 400.2], [-30, -20, 90, 200], [500, 300, 700, 520], [300, 200, 310, 215]],
                     np.float32)
    c, s, mats = T.topdown_affine_params(boxes, (192, 256))
    mats[2] = T.get_udp_warp_matrix(c[2], s[2], 25.0, (192, 256))  # one rotated box
    crops = T.warp_affine_crops(torch.from_numpy(img).cuda(), mats, (192, 256)).cpu().numpy()
    assert crops.shape == (5, 3, 256, 192) and crops.dtype == np.uint8
    for i in range(5):
        ref = warp_ref.warp_affine_u8(img, mats[i], (192, 256)).transpose(2, 0, 1)
        assert np.array_equal(crops[i], ref), i


@pytest.mark.gpu
def test_inference_topdown_end_to_end():
    """image + boxes -> PoseDataSamples; equals pack_crops + test_step on the same crops; keypoints land inside the
    padded box in image coordinates; default box = whole image."""
    import torch

    from probpose_code_amd import apis, synthetic as S
    from probpose_code_amd import transforms as T

    cfg = os.path.join(os.path.dirname(HERE), "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    model = apis.init_model(cfg, dict(state_dict=sd), device="cuda:0", cfg_options={"model.precision": "f32"})
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[100, 50, 220, 400], [300, 100, 520, 460], [10, 10, 60, 90]], np.float32)
    res = apis.inference_topdown(model, img, boxes)
    assert len(res) == 3
    c, s, mats = T.topdown_affine_params(boxes, (192, 256))
    crops = T.warp_affine_crops(torch.from_numpy(img).cuda(), mats, (192, 256))
    ref = model.test_step(apis.pack_crops(crops, c, s, model.dataset_meta, bboxes=boxes))
    for i in range(3):
        kp = res[i].pred_instances.keypoints
        assert kp.shape == (1, 17, 2) and np.array_equal(kp, ref[i].pred_instances.keypoints)
        assert np.allclose(res[i].pred_instances.bboxes, boxes[i][None])
        lo, hi = c[i] - 0.5 * s[i] - 1, c[i] + 0.5 * s[i] + 1
        assert (kp[0] >= lo).all() and (kp[0] <= hi).all()
    whole = apis.inference_topdown(model, img)
    assert len(whole) == 1 and np.allclose(whole[0].pred_instances.bboxes, [[0, 0, 640, 480]])
    xywh = apis.inference_topdown(model, img, np.array([[100, 50, 120, 350]], np.float32), bbox_format="xywh")
    # (not bit-identical: the residual GEMM rotates its K order per workgroup, so the last bit depends on the batch position)
    assert np.allclose(xywh[0].pred_instances.keypoints, res[0].pred_instances.keypoints, atol=1e-3)
    # the multi-person flow (demo/topdown_demo_with_mmdet.py:30-66) with a stand-in detector: category filter, score
    # threshold, box NMS (a near-duplicate of box 0 goes), one batched pass, merged sample
    det = lambda im: (np.concatenate([boxes, boxes[:1] + 2.0, boxes[1:2]]),                     # noqa: E731
                      np.array([0.9, 0.8, 0.7, 0.6, 0.95], np.float32), np.array([0, 0, 0, 0, 1]))
    merged = apis.process_one_image(img, det, model, det_cat_id=0, bbox_thr=0.65, nms_thr=0.3)
    pi = merged.pred_instances
    assert pi.keypoints.shape == (3, 17, 2) and pi.bboxes.shape == (3, 4)
    assert np.allclose(pi.bboxes, boxes) and np.allclose(pi.keypoints, np.concatenate([r.pred_instances.keypoints for r in res]), atol=1e-3)
    assert apis.process_one_image(img, det, model, det_cat_id=2) is None


def test_merge_data_samples_and_image_loading(tmp_path):
    """merge_data_samples (structures/utils.py:16-47) on the shim containers; load_image_bgr returns cv2-order channels."""
    from PIL import Image

    from probpose_code_amd import apis
    from probpose_code_amd.structures import InstanceData, PoseDataSample, merge_data_samples

    samples = []
    for i in range(3):
        ds = PoseDataSample(metainfo=dict(input_center=np.array([i, i], np.float32), input_scale=np.array([2 * i, 3 * i], np.float32),
                                          ori_shape=(10, 20)))
        pi = InstanceData()
        pi.keypoints = np.full((1, 17, 2), float(i))
        pi.keypoint_scores = np.full((1, 17), float(i), np.float32)
        ds.pred_instances = pi
        samples.append(ds)
    m = merge_data_samples(samples)
    assert m.pred_instances.keypoints.shape == (3, 17, 2) and m.pred_instances.keypoints[2, 0, 0] == 2.0
    assert m.input_center.shape == (3, 2) and m.ori_shape == (10, 20)
    with pytest.raises(ValueError):
        merge_data_samples([1, 2])
    rgb = np.zeros((4, 5, 3), np.uint8)
    rgb[..., 0] = 200  # red
    path = str(tmp_path / "red.png")
    Image.fromarray(rgb).save(path)
    bgr = apis.load_image_bgr(path)
    assert bgr.shape == (4, 5, 3) and bgr[0, 0].tolist() == [0, 0, 200]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3", "bf16"])
def test_inference_topdown_stream_equals_frame_by_frame(precision):
    """The video loop with two frames in flight (apis.inference_topdown_stream -> TopdownPoseEstimator.test_step_stream ->
    pipeline.StepPipeline, eager launches, a different number of persons per frame): every field of every pred_instances
    must equal what inference_topdown returns for that frame alone, bit for bit, in frame order."""
    from probpose_code_amd import apis, synthetic as S

    cfg = os.path.join(os.path.dirname(HERE), "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    model = apis.init_model(cfg, dict(state_dict=sd), device="cuda:0", cfg_options={"model.precision": precision})
    rng = np.random.default_rng(17)
    frames = []
    for n in (3, 1, 5, 0, 2, 4, 3):  # persons per frame (0 = no boxes: the whole image)
        img = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
        x0, y0 = rng.uniform(0, 200, n), rng.uniform(0, 120, n)
        boxes = np.stack([x0, y0, x0 + rng.uniform(30, 110, n), y0 + rng.uniform(60, 110, n)], 1).astype(np.float32)
        frames.append((img, boxes if n else None))
    want = [apis.inference_topdown(model, img, bb) for img, bb in frames]
    got = list(apis.inference_topdown_stream(model, iter(frames), depth=2, max_persons=8))
    assert len(got) == len(want)
    fields = ("keypoints", "keypoint_scores", "keypoints_conf", "keypoints_probs", "keypoints_visible", "keypoints_oks",
              "keypoints_error", "bboxes", "bbox_scores")
    for f, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w) == max(1, 0 if frames[f][1] is None else len(frames[f][1]))
        for a, b in zip(g, w):
            for name in fields:
                assert np.array_equal(getattr(a.pred_instances, name), getattr(b.pred_instances, name)), (f, name)
                assert getattr(a.pred_instances, name).dtype == getattr(b.pred_instances, name).dtype, (f, name)
    with pytest.raises(ValueError, match="exceeds max_batch"):
        list(apis.inference_topdown_stream(model, iter(frames[2:3]), max_persons=4))


# ------------------------------------------------------------------------------------- config-driven val pipeline (round 3)
VP = np.load(os.path.join(HERE, "golden", "val_pipeline_cases.npz"))


def test_registered_transforms_and_compose_from_the_config():
    """The val pipeline is composed from cfg.test_dataloader.dataset.pipeline through the TRANSFORMS registry, under the
    reference's type names (mmpose/apis/inference.py:159)."""
    import probpose_code_amd as pp
    from probpose_code_amd import transforms as T
    from probpose_code_amd.config import Config

    for name in ("LoadImage", "GetBBoxCenterScale", "TopdownAffine", "PackPoseInputs"):
        assert pp.TRANSFORMS.get(name) is getattr(T, name) and pp.TRANSFORMS.get("MI355X" + name) is getattr(T, name)
    cfg = Config.fromfile(os.path.join(os.path.dirname(HERE), "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py"))
    pipe = T.Compose(cfg.test_dataloader["dataset"]["pipeline"])
    kinds = [type(t).__name__ for t in pipe.transforms]
    assert kinds == ["LoadImage", "GetBBoxCenterScale", "TopdownAffine", "PackPoseInputs"]
    ta = pipe.transforms[2]
    assert ta.use_udp is True and tuple(ta.input_size) == (192, 256) and ta.input_padding == 1.25
    with pytest.raises(KeyError):
        T.Compose([dict(type="NoSuchTransform")])
    with pytest.raises(TypeError):
        T.Compose([3])


@pytest.mark.parametrize("tag", ["udp1_g1.25_i1.25", "udp1_g1.0_i1.1", "udp0_g1.25_i1.25", "udp0_g1.0_i1.1"])
def test_get_bbox_center_scale_and_topdown_affine_match_the_reference_transform(tag):
    """GetBBoxCenterScale + TopdownAffine.prepare (everything but the image warp) against the REFERENCE's TopdownAffine.transform
    run behind a cv2 stub (tests/golden/make_golden_pipeline.py): centre / scale re-derived from bbox_xyxy_wrt_input with
    input_padding (GetBBoxCenterScale's padding has no effect on the warp - reference behaviour), aspect fix, UDP and
    three-point matrices (use_udp=False), the transformed keypoints and the box in input space."""
    from probpose_code_amd import transforms as T

    udp = tag.startswith("udp1")
    pad_g, pad_i = float(tag.split("_g")[1].split("_i")[0]), float(tag.split("_i")[1])
    g = T.GetBBoxCenterScale(padding=pad_g)
    t = T.TopdownAffine(input_size=(192, 256), input_padding=pad_i, use_udp=udp)
    for i, box in enumerate(VP["boxes"]):
        res = dict(img=None, bbox=box[None].copy(), bbox_score=np.ones(1, np.float32), keypoints=VP["keypoints"][i].copy())
        res = g(res)
        mat = t.prepare(res)
        tol = dict(rtol=0, atol=0) if udp else dict(rtol=1e-6, atol=1e-4)  # three-point solve: cv2's LU vs numpy's, float64
        assert np.allclose(mat, VP[f"{tag}/warp_mat"][i], **tol), i
        for k in ("bbox_center", "bbox_scale", "input_center", "input_scale"):
            assert np.array_equal(np.asarray(res[k]), VP[f"{tag}/{k}"][i]), (k, i)
        assert res["input_size"] == (192, 256)
        assert np.allclose(res["bbox_xyxy_wrt_input"], VP[f"{tag}/bbox_xyxy_wrt_input"][i], rtol=1e-6, atol=2e-3)
        assert np.allclose(res["transformed_keypoints"], VP[f"{tag}/transformed_keypoints"][i], rtol=1e-6, atol=2e-3)


def test_load_image_pad_to_aspect_ratio_and_pack_pose_inputs_cpu(tmp_path):
    """LoadImage: passthrough of an array, file loading with the reference's error text, pad_to_aspect_ratio (255 border so
    that the padded 3:4 box fits, box shifted); PackPoseInputs: gt_instances / metainfo keys of the test path."""
    from PIL import Image

    from probpose_code_amd import transforms as T

    img = np.arange(40 * 60 * 3, dtype=np.uint8).reshape(40, 60, 3)
    res = T.LoadImage()(dict(img=img, bbox=np.array([[5, 5, 30, 30]], np.float32)))
    assert res["img"] is img and res["img_shape"] == (40, 60) and res["ori_shape"] == (40, 60) and res["img_path"] is None
    path = str(tmp_path / "x.png")
    Image.fromarray(img[:, :, ::-1]).save(path)
    res = T.LoadImage()(dict(img_path=path))
    assert np.array_equal(res["img"], img) and res["img_shape"] == (40, 60)
    with pytest.raises(Exception, match="occurs when loading"):
        T.LoadImage()(dict(img_path=str(tmp_path / "missing.png")))
    box = np.array([[40.0, 10.0, 58.0, 38.0]], np.float32)  # 3:4 box padded by 1.25 sticks out on the right / bottom / top
    res = T.LoadImage(pad_to_aspect_ratio=True)(dict(img=img, bbox=box.copy()))
    a = T.fix_bbox_aspect_ratio_xyxy(box, 3 / 4, 1.25).flatten()
    xp = [int(max(0, -a[0])), int(max(0, a[2] - 60))]
    yp = [int(max(0, -a[1])), int(max(0, a[3] - 40))]
    assert res["img"].shape == (40 + yp[0] + yp[1], 60 + xp[0] + xp[1], 3) and res["img_shape"] == res["img"].shape[:2]
    assert np.array_equal(res["img"][yp[0]:yp[0] + 40, xp[0]:xp[0] + 60], img) and (res["img"][:yp[0]] == 255).all()
    assert np.allclose(res["bbox"], box + np.array([xp[0], yp[0], xp[0], yp[0]]))
    packed = T.PackPoseInputs()(dict(img=np.zeros((256, 192, 3), np.uint8), bbox=box, bbox_score=np.ones(1, np.float32),
                                     bbox_scale=np.ones((1, 2), np.float32), input_size=(192, 256), input_center=np.zeros(2), input_scale=np.ones(2),
                                     flip_indices=[0, 2, 1], ori_shape=(40, 60), img_shape=(40, 60), not_a_meta_key=1))
    assert tuple(packed["inputs"].shape) == (3, 256, 192)
    ds = packed["data_samples"]
    assert np.array_equal(ds.gt_instances.bboxes, box) and "bbox_scales" in ds.gt_instances and "bbox_scores" in ds.gt_instances
    assert ds.metainfo["input_size"] == (192, 256) and ds.metainfo["flip_indices"] == [0, 2, 1] and "not_a_meta_key" not in ds.metainfo


@pytest.mark.gpu
def test_inference_topdown_runs_the_configs_pipeline_and_the_demo_script(tmp_path):
    """BASELINE config 1: demo/image_demo.py as a subprocess (image file + config + synthetic checkpoint) must print what
    inference_topdown returns in-process; and the pipeline switches of the config reach the warp: use_udp=False and another
    input_padding change the crop exactly as the oracle warp with the reference's matrix says."""
    import json
    import subprocess
    import sys

    import torch

    from oracle import warp_ref
    from probpose_code_amd import apis, synthetic
    from probpose_code_amd import transforms as T

    root = os.path.dirname(HERE)
    cfg = os.path.join(root, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
    img_path = os.path.join(root, "demo", "resources", "synthetic_person.png")
    out_file = str(tmp_path / "out.json")
    boxes = "40,30,200,400;100,60,300,420"
    r = subprocess.run([sys.executable, os.path.join(root, "demo", "image_demo.py"), img_path, cfg, "synthetic", "--out-file", out_file,
                        "--bboxes", boxes], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out_file))
    ckpt = dict(state_dict=synthetic.synthetic_state_dict("small", seed=0, logit_scale=2.0))
    model = apis.init_model(cfg, ckpt, device="cuda:0")
    assert model.engine.precision == "f16x3"  # the config's default is the mode within 1e-3
    bb = np.array([[float(v) for v in b.split(",")] for b in boxes.split(";")], np.float32)
    res = apis.inference_topdown(model, img_path, bb)
    assert len(res) == len(got) == 2
    for g, s in zip(got, res):
        pi = s.pred_instances
        assert np.array_equal(np.asarray(g["keypoints"]), pi.keypoints[0]) and np.array_equal(np.asarray(g["keypoint_scores"], np.float32), pi.keypoint_scores[0])
        assert np.allclose(g["bbox"], pi.bboxes[0])
    # pipeline switches: the crop the model sees
    img = apis.load_image_bgr(img_path)
    for udp, pad in ((False, 1.25), (True, 1.1)):
        pipe = T.Compose([dict(type="LoadImage"), dict(type="GetBBoxCenterScale"),
                          dict(type="TopdownAffine", input_size=(192, 256), use_udp=udp, input_padding=pad), dict(type="PackPoseInputs")])
        packed = pipe.batched([dict(img=img, bbox=b[None].copy(), bbox_score=np.ones(1, np.float32)) for b in bb])
        for b, pk in zip(bb, packed):
            c, s = T.bbox_xyxy2cs(b, padding=pad)
            s = T.fix_aspect_ratio(s.reshape(1, 2), 192 / 256)[0]
            m = T.get_udp_warp_matrix(c, s, 0.0, (192, 256)) if udp else T.get_warp_matrix(c, s, 0.0, (192, 256))
            want = warp_ref.warp_affine_u8(img, np.asarray(m, np.float64), (192, 256))
            assert pk["inputs"].is_cuda and np.array_equal(pk["inputs"].permute(1, 2, 0).cpu().numpy(), want)
            assert np.allclose(pk["data_samples"].metainfo["input_scale"], s)
    # single-sample path == batched path
    one = pipe(dict(img=img, bbox=bb[0][None].copy(), bbox_score=np.ones(1, np.float32)))
    assert torch.equal(one["inputs"], packed[0]["inputs"])


@pytest.mark.gpu
def test_warp_differential_fuzz_against_the_oracle():
    """tests/fuzz_warp.py for a few seconds: images down to 1 x 3 pixels, boxes across / outside the image, slivers, huge boxes, rotations, both input
    sizes - the HIP crops equal the oracle's (cv2.warpAffine's fixed-point bilinear path) byte for byte."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_warp.py"), "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WARP FUZZ OK" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
