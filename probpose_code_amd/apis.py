"""Caller-side helpers mirroring ``mmpose/apis/inference.py``: ``init_model`` (:66-130) and the packing of
already-cropped inputs into the ``{inputs, data_samples}`` batch that ``model.test_step`` consumes
(what ``inference_topdown`` :133-200 builds through the val pipeline + ``pseudo_collate``).

The crop pipeline itself (LoadImage / GetBBoxCenterScale / TopdownAffine with cv2.warpAffine) is the
"next" row of SURVEY.md 8f and not part of this round: callers hand over 256x192 uint8 crops plus the
``input_center`` / ``input_scale`` that ``TopdownAffine`` recorded for them.
"""
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .config import Config
from .pose_estimators import build_pose_estimator
from .structures import InstanceData, PoseDataSample
from .synthetic import COCO_FLIP_INDICES


def coco_dataset_meta() -> dict:
    """The slice of ``parse_pose_metainfo(configs/_base_/datasets/coco.py)`` the path reads
    (mmpose/datasets/datasets/utils.py:155-190): flip_indices derived from the `swap` pairs."""
    return dict(dataset_name="coco", num_keypoints=17, flip_indices=list(COCO_FLIP_INDICES))


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
        ckpt = torch.load(checkpoint, map_location="cpu") if isinstance(checkpoint, str) else checkpoint
        sd = ckpt.get("state_dict", ckpt)
        model.load_state_dict(sd, strict=False)
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
