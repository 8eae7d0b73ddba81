"""Caller-side helpers mirroring ``mmpose/apis/inference.py``: ``init_model`` (:66-130) and the packing of
already-cropped inputs into the ``{inputs, data_samples}`` batch that ``model.test_step`` consumes
(what ``inference_topdown`` :133-200 builds through the val pipeline + ``pseudo_collate``).

``inference_topdown`` (:133-200) runs the val pipeline of the config on the device: GetBBoxCenterScale and the
TopdownAffine box arithmetic on the host (``transforms.py``), the image warp as one HIP launch for all boxes
(``pp_warp_affine_u8``), then ``test_step``. Callers that already hold 256x192 uint8 crops use ``pack_crops``.
"""
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .config import Config
from .pose_estimators import build_pose_estimator
from .structures import InstanceData, PoseDataSample
from .synthetic import COCO_FLIP_INDICES
from . import transforms as T


def coco_dataset_meta() -> dict:
    """The slice of ``parse_pose_metainfo(configs/_base_/datasets/coco.py)`` the path reads
    (mmpose/datasets/datasets/utils.py:155-190): flip_indices derived from the `swap` pairs."""
    return dict(dataset_name="coco", num_keypoints=17, flip_indices=list(COCO_FLIP_INDICES))


def load_state_dict_checked(model, state_dict: dict) -> None:
    """``load_state_dict`` through the model's pre-hooks (base.py:212-243, probmap_head.py:1014-1061), then account for
    every key. mmengine's ``load_checkpoint`` loads non-strictly and LOGS mismatches; here a checkpoint that leaves any
    parameter of the model unset is an error (the model would silently run on its random init), and keys the model
    does not know are reported with a warning, as mmengine does."""
    import warnings

    res = model.load_state_dict(dict(state_dict), strict=False)
    missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")]
    if missing:
        raise RuntimeError(f"checkpoint does not provide {len(missing)} parameter(s) of the model, e.g. {missing[:8]} "
                           "(key names must be the reference's: backbone.* as mmpretrain's VisionTransformer, head.* as ProbMapHead)")
    if res.unexpected_keys:
        warnings.warn(f"unexpected key(s) in the checkpoint's state_dict, ignored: {list(res.unexpected_keys)[:8]}"
                      f"{' ...' if len(res.unexpected_keys) > 8 else ''}", RuntimeWarning, stacklevel=2)


def init_model(config: Union[str, Config, dict], checkpoint: Optional[Union[str, dict]] = None, device: str = "cuda:0",
               cfg_options: Optional[dict] = None):
    """apis/inference.py:66-130. ``checkpoint`` may be a path (torch.load) or a state dict."""
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif isinstance(config, dict):
        config = Config(config)
    elif not isinstance(config, Config):
        raise TypeError(f"config must be a filename or Config object, but got {type(config)}")
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    model_cfg = dict(config.model)
    model_cfg.pop("train_cfg", None)
    model = build_pose_estimator(model_cfg)
    dataset_meta = None
    if checkpoint is not None:
        # mmengine's load_checkpoint (apis/inference.py:103) unpickles the whole file: meta holds numpy arrays
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False) if isinstance(checkpoint, str) else checkpoint
        sd = ckpt.get("state_dict", ckpt)
        load_state_dict_checked(model, sd)
        dataset_meta = ckpt.get("meta", {}).get("dataset_meta") if isinstance(ckpt.get("meta", None), dict) else None
    model.dataset_meta = dataset_meta or coco_dataset_meta()
    model.cfg = config
    model.to(device)
    model.eval()
    return model


def pack_crops(crops_u8: torch.Tensor, input_center: np.ndarray, input_scale: np.ndarray, dataset_meta: dict,
               bboxes: Optional[np.ndarray] = None, bbox_scores: Optional[np.ndarray] = None) -> dict:
    """(B,3,H,W) uint8 crops + TopdownAffine metadata -> the batch dict of ``pseudo_collate``
    (``inputs``: list of uint8 CHW tensors, ``data_samples``: list of PoseDataSample with the meta keys
    PackPoseInputs forwards, mmpose/datasets/transforms/formatting.py:179-277)."""
    B, _, H, W = crops_u8.shape
    samples = []
    for b in range(B):
        ds = PoseDataSample()
        gt = InstanceData()
        if bboxes is not None:
            gt.bboxes = np.asarray(bboxes[b], np.float32).reshape(1, 4)
        else:
            c, s = np.asarray(input_center[b], np.float32), np.asarray(input_scale[b], np.float32)
            gt.bboxes = np.concatenate([c - 0.5 * s / 1.25, c + 0.5 * s / 1.25]).reshape(1, 4).astype(np.float32)
        gt.bbox_scores = np.ones(1, np.float32) if bbox_scores is None else np.asarray(bbox_scores[b], np.float32).reshape(1)
        ds.gt_instances = gt
        ds.set_metainfo(dict(
            input_size=(W, H), input_center=np.asarray(input_center[b], np.float32),
            input_scale=np.asarray(input_scale[b], np.float32), flip_indices=list(dataset_meta["flip_indices"]),
            dataset_name=dataset_meta.get("dataset_name", "coco"), img_shape=(H, W), ori_shape=(H, W),
        ))
        samples.append(ds)
    return dict(inputs=[crops_u8[b] for b in range(B)], data_samples=samples)


def load_image_bgr(path: str) -> np.ndarray:
    """LoadImage for a file path (mmpose/datasets/transforms/loading.py:47-107 -> mmcv.imread, BGR uint8). mmcv decodes
    with cv2 (flag 'color': IMREAD_COLOR, EXIF orientation ignored by `imdecode`): when cv2 is importable the same call is
    used, so the crops see the very bytes the reference sees. Without cv2 (this image) Pillow decodes; its JPEG decoder
    may differ from cv2's bundled libjpeg-turbo by one grey level in a few pixels."""
    try:
        import cv2  # type: ignore

        img = cv2.imdecode(np.fromfile(path, dtype=np.uint8), cv2.IMREAD_COLOR)
        if img is None:
            raise OSError(f"cv2 could not decode {path}")
        return img
    except ImportError:
        pass
    from PIL import Image

    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def inference_topdown(model, img: Union[np.ndarray, torch.Tensor], bboxes=None, bbox_format: str = "xyxy"):
    """apis/inference.py:133-200 for an image already in memory: ``img`` is (H, W, 3) uint8 BGR (what LoadImage /
    cv2.imread produce), host array or device tensor; ``bboxes`` (N, 4) in ``bbox_format``, None or empty = the whole
    image as one box; ``img`` may also be a file path. Returns list[PoseDataSample], one per box, keypoints in image
    coordinates."""
    batch = _frame_batch(model, img, bboxes, bbox_format)
    with torch.no_grad():
        return model.test_step(batch)


DEFAULT_VAL_PIPELINE = [
    dict(type="LoadImage"),
    dict(type="GetBBoxCenterScale"),
    dict(type="TopdownAffine", input_size=(192, 256), use_udp=True),
    dict(type="PackPoseInputs"),
]


def _val_pipeline(model) -> T.Compose:
    """``Compose(model.cfg.test_dataloader.dataset.pipeline)`` (apis/inference.py:159), built once per model. Configs without a
    test dataloader (a bare model dict) get the ProbPose val pipeline at the decoder's input size."""
    pipe = getattr(model, "_val_pipeline", None)
    if pipe is None:
        cfg = getattr(model, "cfg", None)
        steps = None
        if cfg is not None:
            try:
                steps = cfg["test_dataloader"]["dataset"]["pipeline"]
            except (KeyError, TypeError):
                steps = cfg.get("val_pipeline", None) if hasattr(cfg, "get") else None
        if not steps:
            steps = [dict(t) for t in DEFAULT_VAL_PIPELINE]
            steps[2]["input_size"] = tuple(int(v) for v in model.head.decoder.input_size)
        pipe = T.Compose([dict(t) for t in steps])
        for t in pipe.transforms:
            if isinstance(t, T.TopdownAffine) and t.device is None:
                t.device = str(next(model.parameters()).device)
        model._val_pipeline = pipe
    return pipe


def _frame_batch(model, img, bboxes, bbox_format):
    """(image, boxes) -> the ``test_step`` batch dict of ``inference_topdown``: one ``data_info`` per box through the config's
    val pipeline (box arithmetic on the host, ONE warp launch for all boxes of the image), then ``pseudo_collate``."""
    pipeline = _val_pipeline(model)
    img_path = img if isinstance(img, str) else None  # kept beside the decoded-once pixels (apis/inference.py:182-186: dict(img_path=img))
    if bboxes is None or len(bboxes) == 0:
        if isinstance(img, str):
            img = load_image_bgr(img)  # (the reference opens the file for its size, then LoadImage reads it again per box)
        h, w = img.shape[:2]
        bboxes = np.array([[0, 0, w, h]], dtype=np.float32)
    else:
        bboxes = np.array(bboxes) if isinstance(bboxes, list) else np.asarray(bboxes)
        assert bbox_format in {"xyxy", "xywh"}, f'Invalid bbox_format "{bbox_format}".'
        if bbox_format == "xywh":
            bboxes = T.bbox_xywh2xyxy(bboxes)
        if isinstance(img, str):
            img = load_image_bgr(img)  # decoded once for all boxes
    data_list = []
    for bbox in bboxes:
        data_info = dict(img=img)
        if img_path is not None:
            data_info["img_path"] = img_path  # LoadImage keeps an existing img_path; PackPoseInputs forwards it as metainfo
        data_info["bbox"] = bbox[None, :4]  # shape (1, 4), dtype as given (:185; a trailing score column is dropped, the reference's bbox_xyxy2cs would choke on it)
        data_info["bbox_score"] = np.ones(1, dtype=np.float32)  # shape (1,)
        data_info.update(model.dataset_meta)
        data_list.append(data_info)
    return T.pseudo_collate(pipeline.batched(data_list))


def inference_topdown_stream(model, frames, bbox_format: str = "xyxy", depth: int = 2, max_persons: int = 64):
    """``inference_topdown`` over a sequence of frames - the video loop of demo/topdown_demo_with_mmdet.py:280-310
    (``while cap.isOpened(): ... process_one_image(...)``) - with up to ``depth`` frames in flight on the device: ``frames``
    yields ``(img, bboxes)`` pairs (``bboxes`` None / empty = the whole image), the generator yields one
    ``list[PoseDataSample]`` per frame, in order, identical to what ``inference_topdown`` returns for that frame."""
    with torch.no_grad():
        yield from model.test_step_stream((_frame_batch(model, img, bb, bbox_format) for img, bb in frames), depth=depth,
                                          max_batch=max_persons)


def process_one_image(img, detector, pose_estimator, det_cat_id: int = 0, bbox_thr: float = 0.3, nms_thr: float = 0.3):
    """The multi-person flow of ``demo/topdown_demo_with_mmdet.py:30-66`` without the drawing: detector -> boxes of category
    ``det_cat_id`` above ``bbox_thr`` -> box NMS -> ``inference_topdown`` (all persons of the frame in one batch: one warp
    launch, one pass of the hot path) -> ``merge_data_samples``. ``detector`` is any callable ``img -> (bboxes (N, 4) xyxy,
    scores (N,), labels (N,))`` - with mmdet installed: ``lambda im: (lambda r: (r.bboxes, r.scores, r.labels))(
    inference_detector(det_model, im).pred_instances.cpu().numpy())``. Returns the merged PoseDataSample (its
    ``pred_instances`` hold every person; ``None`` when nothing was detected, as the reference returns)."""
    from .evaluation import nms
    from .structures import merge_data_samples

    if isinstance(img, str):
        img = load_image_bgr(img)
    boxes, scores, labels = (np.asarray(a) for a in detector(img))
    dets = np.concatenate((boxes.reshape(-1, 4), scores.reshape(-1, 1)), axis=1)
    dets = dets[np.logical_and(labels.reshape(-1) == det_cat_id, scores.reshape(-1) > bbox_thr)]
    dets = dets[nms(dets, nms_thr), :4]
    if len(dets) == 0:
        return None
    return merge_data_samples(inference_topdown(pose_estimator, img, dets))
