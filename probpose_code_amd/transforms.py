"""The val-pipeline transforms in front of the hot path, for ``apis.inference_topdown``:
``LoadImage`` -> ``GetBBoxCenterScale`` -> ``TopdownAffine`` -> ``PackPoseInputs``, registered in ``TRANSFORMS`` under the
reference's names and composed from ``cfg.test_dataloader.dataset.pipeline`` as the reference does
(mmpose/apis/inference.py:159). The box arithmetic is host numpy exactly as in the reference (it is a handful of scalars per
person); the image warp runs on the device (``pp_warp_affine_u8``), one launch for all boxes of an image when the pipeline
is applied to a list (``Compose.batched``).
"""
import math
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .registry import OVERRIDE_REFERENCE_NAMES, TRANSFORMS


def bbox_xywh2xyxy(bbox: np.ndarray) -> np.ndarray:
    """mmpose/structures/bbox/transforms.py:12-26."""
    bbox = np.array(bbox, copy=True)
    bbox[..., 2:4] = bbox[..., 2:4] + bbox[..., 0:2]
    return bbox


def bbox_xyxy2cs(bbox: np.ndarray, padding: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """(left, top, right, bottom) -> (center, scale * padding); mmpose/structures/bbox/transforms.py:44-72."""
    dim = bbox.ndim
    if dim == 1:
        bbox = bbox[None, :]
    scale = (bbox[..., 2:] - bbox[..., :2]) * padding
    center = (bbox[..., 2:] + bbox[..., :2]) * 0.5
    if dim == 1:
        center, scale = center[0], scale[0]
    return center, scale


def fix_aspect_ratio(bbox_scale: np.ndarray, aspect_ratio: float) -> np.ndarray:
    """TopdownAffine._fix_aspect_ratio (mmpose/datasets/transforms/topdown_transforms.py:50-68): (n, 2) scales grown to w/h."""
    w, h = np.hsplit(bbox_scale, [1])
    return np.where(w > h * aspect_ratio, np.hstack([w, w / aspect_ratio]), np.hstack([h * aspect_ratio, h]))


def get_udp_warp_matrix(center: np.ndarray, scale: np.ndarray, rot: float, output_size: Tuple[int, int]) -> np.ndarray:
    """2x3 float32 matrix of the unbiased-data-processing warp; mmpose/structures/bbox/transforms.py:315-359."""
    assert len(center) == 2 and len(scale) == 2 and len(output_size) == 2
    input_size = center * 2
    rot_rad = np.deg2rad(rot)
    m = np.zeros((2, 3), dtype=np.float32)
    sx = (output_size[0] - 1) / scale[0]
    sy = (output_size[1] - 1) / scale[1]
    # math.cos / math.sin as in the reference: Python floats are "weak" in NumPy's promotion, so with float32 centres
    # and scales the products below stay float32 - np.cos would silently turn them into float64 and change the last bit
    c, s = math.cos(rot_rad), math.sin(rot_rad)
    m[0, 0] = c * sx
    m[0, 1] = -s * sx
    m[0, 2] = sx * (-0.5 * input_size[0] * c + 0.5 * input_size[1] * s + 0.5 * scale[0])
    m[1, 0] = s * sy
    m[1, 1] = c * sy
    m[1, 2] = sy * (-0.5 * input_size[0] * s - 0.5 * input_size[1] * c + 0.5 * scale[1])
    return m


def invert_affine(m: np.ndarray) -> np.ndarray:
    """The dst -> src map cv2.warpAffine derives from M when WARP_INVERSE_MAP is not set (imgwarp.cpp), float64."""
    M = np.asarray(m, np.float64).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def get_warp_matrix(center, scale, rot, output_size, inv=False):
    """mmpose/structures/bbox/transforms.py:362-426 (shift 0, fix_aspect_ratio=True): the box of size ``scale`` around
    ``center``, rotated by ``rot`` degrees, onto ``output_size`` = (w, h). Three point pairs held in float32 like the
    reference's (centre; the point half a box width to its left, rotated; the third perpendicular to those two, computed
    in float32), then the affine map through them - cv2.getAffineTransform in the reference, here the 3x3 solve
    [x y 1] A = [x' y'] in float64."""
    center, scale = np.asarray(center, np.float64).reshape(2), np.asarray(scale, np.float64).reshape(2)
    rad = np.deg2rad(rot)
    sn, cs = np.sin(rad), np.cos(rad)
    half = scale[0] * -0.5
    src, dst = np.ones((3, 3), np.float32), np.ones((3, 3), np.float32)  # homogeneous rows
    src[0, :2], src[1, :2] = center, center + np.array([cs * half - sn * 0.0, sn * half + cs * 0.0])
    dst[0, :2], dst[1, :2] = [output_size[0] * 0.5, output_size[1] * 0.5], [0.0, output_size[1] * 0.5]
    for p in (src, dst):
        p[2, 0], p[2, 1] = p[1, 0] - (p[0, 1] - p[1, 1]), p[1, 1] + (p[0, 0] - p[1, 0])
    a, b = (dst, src) if inv else (src, dst)
    return np.linalg.solve(a.astype(np.float64), b[:, :2].astype(np.float64)).T


def topdown_affine_params(bboxes_xyxy: np.ndarray, input_size: Tuple[int, int], padding: float = 1.25, input_padding: float = 1.25):
    """What GetBBoxCenterScale + TopdownAffine compute per box before the warp (common_transforms.py:57-85,
    topdown_transforms.py:83-117): returns (centers (n,2), scales (n,2), warp matrices (n,2,3) float32). As in the
    reference the scale comes from ``bbox_xyxy_wrt_input`` with ``input_padding`` and is then fixed to the w/h ratio."""
    w, h = input_size
    n = len(bboxes_xyxy)
    centers, scales, mats = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 2, 3), np.float32)
    for i, bb in enumerate(np.asarray(bboxes_xyxy, np.float32)):
        c, s = bbox_xyxy2cs(bb, padding=input_padding)
        s = fix_aspect_ratio(s.reshape(1, 2), aspect_ratio=w / h)[0]
        centers[i], scales[i] = c, s
        mats[i] = get_udp_warp_matrix(c, s, 0.0, output_size=(w, h))
    return centers, scales, mats


def warp_affine_crops(img: torch.Tensor, mats: np.ndarray, input_size: Tuple[int, int]) -> torch.Tensor:
    """img: (H, W, C) uint8 device tensor (BGR as cv2.imread gives it); mats: (n, 2, 3) forward warp matrices.
    Returns (n, C, h, w) uint8 crops on the device."""
    if not img.is_cuda:
        raise RuntimeError("probpose_code_amd.transforms.warp_affine_crops runs on the GPU only (no CPU fallback)")
    assert img.dtype == torch.uint8 and img.dim() == 3
    img = img.contiguous()
    w, h = int(input_size[0]), int(input_size[1])
    n = len(mats)
    inv = torch.as_tensor(np.stack([invert_affine(m) for m in mats]) if n else np.zeros((0, 2, 3)), dtype=torch.float64).to(img.device)
    out = torch.empty((n, img.shape[2], h, w), dtype=torch.uint8, device=img.device)
    _lib.call("pp_warp_affine_u8", img.data_ptr(), img.shape[0], img.shape[1], img.shape[2], inv.data_ptr(), out.data_ptr(), n, h, w,
              torch.cuda.current_stream(img.device).cuda_stream)
    return out


# ------------------------------------------------------------------------------------------------- registered transforms
def _register(name):
    """Under the reference's name when this package owns the registry (or was asked to override a real MMPose), always under
    ``MI355X<name>``."""

    def deco(cls):
        TRANSFORMS.register_module(name="MI355X" + name, force=True, module=cls)
        if OVERRIDE_REFERENCE_NAMES:
            TRANSFORMS.register_module(name=name, force=True, module=cls)
        return cls

    return deco


class BaseTransform:
    """mmcv.transforms.BaseTransform [3P]: ``__call__`` = ``transform``."""

    def __call__(self, results: Dict) -> Optional[Dict]:
        return self.transform(results)


def fix_bbox_aspect_ratio_xyxy(bbox: np.ndarray, aspect_ratio: float = 3 / 4, padding: float = 1.25) -> np.ndarray:
    """mmpose/structures/keypoint/keypoints_min_padding.py:68-133 for ``bbox_format="xyxy"``: every box grown around its
    centre to w / h = ``aspect_ratio`` (a zero width / height counts as 1 in the comparison), then padded."""
    shape = bbox.shape
    b = np.array(bbox).reshape(-1, 4)
    centers = b[:, :2] + (b[:, 2:] - b[:, :2]) / 2
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    nw, nh = w.copy().astype(np.float32), h.copy().astype(np.float32)
    wc, hc = np.where(w == 0, 1, w), np.where(h == 0, 1, h)
    wide = wc / hc > aspect_ratio
    nh = np.where(wide, wc / aspect_ratio, nh).astype(np.float32)
    nw = np.where(wide, nw, hc * aspect_ratio).astype(np.float32)
    nw, nh = nw * padding, nh * padding
    out = np.array([centers[:, 0] - nw / 2, centers[:, 1] - nh / 2, centers[:, 0] + nw / 2, centers[:, 1] + nh / 2]).T
    return out.reshape(shape)


@_register("LoadImage")
class LoadImage(BaseTransform):
    """mmpose/datasets/transforms/loading.py:12-107: ``results['img']`` from ``img_path`` (BGR uint8; mmcv.imread [3P]) or the
    array / device tensor already there; ``img_shape``, ``ori_shape``; ``pad_to_aspect_ratio``: the image is padded with 255
    so that the padded 3:4 box lies inside it, box (and keypoints, when present) shifted."""

    def __init__(self, pad_to_aspect_ratio: bool = False, to_float32: bool = False, color_type: str = "color",
                 imdecode_backend: str = "cv2", backend_args: Optional[dict] = None, ignore_empty: bool = False, **kwargs):
        if color_type != "color":
            raise NotImplementedError("LoadImage: only color_type='color' (3-channel BGR) is implemented")
        self.pad_to_aspect_ratio, self.to_float32, self.ignore_empty = pad_to_aspect_ratio, to_float32, ignore_empty

    def transform(self, results: dict) -> Optional[dict]:
        try:
            if "img" not in results:
                from .apis import load_image_bgr

                try:
                    img = load_image_bgr(results["img_path"])
                except OSError:
                    if self.ignore_empty:
                        return None
                    raise
                if self.to_float32:
                    img = img.astype(np.float32)
                results["img"] = img
                results["img_shape"] = img.shape[:2]
                results["ori_shape"] = img.shape[:2]
            else:
                img = results["img"]
                assert isinstance(img, (np.ndarray, torch.Tensor))
                if self.to_float32:
                    img = img.astype(np.float32) if isinstance(img, np.ndarray) else img.float()
                if "img_path" not in results:
                    results["img_path"] = None
                results["img_shape"] = tuple(img.shape[:2])
                results["ori_shape"] = tuple(img.shape[:2])
            if self.pad_to_aspect_ratio:
                a = fix_bbox_aspect_ratio_xyxy(np.asarray(results["bbox"]), aspect_ratio=3 / 4, padding=1.25).flatten()
                x_pad = np.array([max(0, -a[0]), max(0, a[2] - results["img_shape"][1])], dtype=int)
                y_pad = np.array([max(0, -a[1]), max(0, a[3] - results["img_shape"][0])], dtype=int)
                img = results["img"]
                if isinstance(img, torch.Tensor):
                    img = torch.nn.functional.pad(img.permute(2, 0, 1), (int(x_pad[0]), int(x_pad[1]), int(y_pad[0]), int(y_pad[1])),
                                                  value=255).permute(1, 2, 0).contiguous()
                else:
                    img = np.pad(img, ((y_pad[0], y_pad[1]), (x_pad[0], x_pad[1]), (0, 0)), mode="constant", constant_values=255)
                results["img"] = img
                bbox = np.array(results["bbox"]).flatten()
                bbox[:2] += np.array([x_pad[0], y_pad[0]])
                bbox[2:] += np.array([x_pad[0], y_pad[0]])
                results["bbox"] = bbox.reshape(np.array(results["bbox"]).shape)
                if results.get("keypoints", None) is not None:  # (the reference indexes results['keypoints'] unconditionally)
                    kpts = np.array(results["keypoints"]).reshape(-1, 2)
                    kpts[:, :2] += np.array([x_pad[0], y_pad[0]])
                    results["keypoints"] = kpts.reshape(np.array(results["keypoints"]).shape)
                results["img_shape"] = tuple(img.shape[:2])
                results["ori_shape"] = tuple(img.shape[:2])
        except Exception as e:  # noqa: BLE001 -- the reference re-raises every failure with the file name (loading.py:100-105)
            raise type(e)(f'`{str(e)}` occurs when loading `{results.get("img_path")}`.Please check whether the file exists.')
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(pad_to_aspect_ratio={self.pad_to_aspect_ratio}, to_float32={self.to_float32})"


@_register("GetBBoxCenterScale")
class GetBBoxCenterScale(BaseTransform):
    """mmpose/datasets/transforms/common_transforms.py:32-92."""

    def __init__(self, padding: float = 1.25) -> None:
        self.padding = padding

    def transform(self, results: Dict) -> Optional[dict]:
        results["bbox_xyxy_wrt_input"] = results["bbox"]  # the original box, TopdownAffine's input
        if "bbox_center" in results and "bbox_scale" in results:
            warnings.warn('Use the existing "bbox_center" and "bbox_scale". The padding will still be applied.')
            results["bbox_scale"] = results["bbox_scale"] * self.padding
        else:
            center, scale = bbox_xyxy2cs(results["bbox"], padding=self.padding)
            results["bbox_center"] = center
            results["bbox_scale"] = scale
        return results

    def __repr__(self) -> str:
        return self.__class__.__name__ + f"(padding={self.padding})"


def _affine_points(pts: np.ndarray, m: np.ndarray) -> np.ndarray:
    """cv2.transform(pts, m) for (..., 2) points and a 2x3 matrix."""
    pts = np.asarray(pts)
    out = pts[..., :2].astype(np.float64) @ np.asarray(m, np.float64)[:, :2].T + np.asarray(m, np.float64)[:, 2]
    return out.astype(pts.dtype if np.issubdtype(pts.dtype, np.floating) else np.float64)


@_register("TopdownAffine")
class TopdownAffine(BaseTransform):
    """mmpose/datasets/transforms/topdown_transforms.py:14-150. Box arithmetic as there (centre / scale re-derived from
    ``bbox_xyxy_wrt_input`` with ``input_padding``, fixed to the input's aspect ratio, UDP or three-point warp matrix);
    the warp itself (cv2.warpAffine INTER_LINEAR there) is ``pp_warp_affine_u8`` on the device and leaves ``results['img']``
    as a (C, h, w) uint8 DEVICE tensor - the layout PackPoseInputs hands to the model. ``bbox_mask`` (a training-loss input,
    warped with the same matrix) is produced only with ``with_bbox_mask=True``: nothing on the test path reads it."""

    def __init__(self, input_size: Tuple[int, int], input_padding: float = 1.25, use_udp: bool = False, with_bbox_mask: bool = False,
                 device: Optional[str] = None) -> None:
        assert len(input_size) == 2 and all(isinstance(v, int) for v in input_size), f"Invalid input_size {input_size}"
        self.input_size, self.use_udp, self.input_padding = tuple(input_size), use_udp, input_padding
        self.with_bbox_mask, self.device = with_bbox_mask, device

    _fix_aspect_ratio = staticmethod(fix_aspect_ratio)

    def prepare(self, results: Dict) -> np.ndarray:
        """Everything but the warp: updates the box fields, returns the 2x3 forward matrix."""
        w, h = self.input_size
        bbox_xyxy_wrt_input = results.get("bbox_xyxy_wrt_input", None)
        if bbox_xyxy_wrt_input is not None:
            _c, _s = bbox_xyxy2cs(bbox_xyxy_wrt_input, padding=self.input_padding)
            results["bbox_center"] = _c.reshape(1, 2)
            results["bbox_scale"] = _s.reshape(1, 2)
        results["bbox_scale"] = self._fix_aspect_ratio(results["bbox_scale"], aspect_ratio=w / h)
        assert results["bbox_center"].shape[0] == 1, (
            "Top-down heatmap only supports single instance. Got invalid "
            f'shape of bbox_center {results["bbox_center"].shape}.')
        center, scale = results["bbox_center"][0], results["bbox_scale"][0]
        rot = results["bbox_rotation"][0] if "bbox_rotation" in results else 0.0
        if self.use_udp:
            warp_mat = get_udp_warp_matrix(center, scale, rot, output_size=(w, h))
        else:
            warp_mat = get_warp_matrix(center, scale, rot, output_size=(w, h)).astype(np.float64)
        if results.get("keypoints", None) is not None:
            src = results["transformed_keypoints"] if results.get("transformed_keypoints", None) is not None else results["keypoints"]
            tk = np.array(src, copy=True)
            tk[..., :2] = _affine_points(tk[..., :2], warp_mat)
            results["transformed_keypoints"] = tk
        if bbox_xyxy_wrt_input is not None:
            bb = np.array(bbox_xyxy_wrt_input, copy=True).reshape(1, 2, 2)
            results["bbox_xyxy_wrt_input"] = _affine_points(bb, warp_mat).reshape(1, 4)
        results["input_size"] = (w, h)
        results["input_center"] = center
        results["input_scale"] = scale
        return warp_mat

    def _device_image(self, img, cache: Optional[dict] = None) -> torch.Tensor:
        if isinstance(img, torch.Tensor) and img.is_cuda:
            return img
        key = id(img)
        if cache is not None and key in cache:
            return cache[key]
        t = (img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))).to(self.device or "cuda")
        if cache is not None:
            cache[key] = t
        return t

    def _bbox_mask(self, results, img_t, warp_mat, box_before):
        img_h, img_w = img_t.shape[:2]
        b = np.array(box_before, copy=True).flatten()
        b[:2] = np.maximum(b[:2], 0)
        b[2:4] = np.minimum(b[2:4], [img_w, img_h])
        x0, y0, x1, y1 = b[:4].astype(int)
        mask = torch.zeros((img_h, img_w, 1), dtype=torch.uint8, device=img_t.device)
        mask[y0:y1, x0:x1] = 1
        results["bbox_mask"] = warp_affine_crops(mask, warp_mat[None], self.input_size)[0].cpu().numpy().reshape(1, self.input_size[1], self.input_size[0])

    def transform(self, results: Dict) -> Optional[dict]:
        box_before = results.get("bbox_xyxy_wrt_input", results.get("bbox"))
        warp_mat = self.prepare(results)
        img_t = self._device_image(results["img"])
        if self.with_bbox_mask:
            self._bbox_mask(results, img_t, warp_mat, box_before)
        results["img"] = warp_affine_crops(img_t, np.asarray(warp_mat, np.float64)[None], self.input_size)[0]
        return results

    def transform_batch(self, results_list: List[Dict]) -> List[Dict]:
        """The same for a list of samples, ONE warp launch per distinct source image (the persons of a frame)."""
        mats, boxes = [], []
        for r in results_list:
            boxes.append(r.get("bbox_xyxy_wrt_input", r.get("bbox")))
            mats.append(np.asarray(self.prepare(r), np.float64))
        cache: dict = {}
        groups: Dict[int, List[int]] = {}
        for i, r in enumerate(results_list):
            groups.setdefault(id(r["img"]), []).append(i)
        for idx in groups.values():
            img_t = self._device_image(results_list[idx[0]]["img"], cache)
            crops = warp_affine_crops(img_t, np.stack([mats[i] for i in idx]), self.input_size)
            for k, i in enumerate(idx):
                if self.with_bbox_mask:
                    self._bbox_mask(results_list[i], img_t, mats[i], boxes[i])
                results_list[i]["img"] = crops[k]
        return results_list

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(input_size={self.input_size}, use_udp={self.use_udp})"


@_register("PackPoseInputs")
class PackPoseInputs(BaseTransform):
    """mmpose/datasets/transforms/formatting.py:61-285 for the keys the test path carries: ``inputs`` = the crop as a
    (C, h, w) tensor (already on the device after TopdownAffine), ``data_samples`` = PoseDataSample with ``gt_instances``
    (bbox -> bboxes, bbox_score -> bbox_scores, bbox_scale -> bbox_scales, keypoints ... as the reference's mapping table)
    and the ``meta_keys`` present in the results."""

    instance_mapping_table = dict(bbox="bboxes", bbox_score="bbox_scores", keypoints="keypoints", keypoints_cam="keypoints_cam",
                                  keypoints_visible="keypoints_visible", keypoints_visibility="keypoints_visibility",
                                  bbox_scale="bbox_scales", head_size="head_size", in_image="in_image", keypoints_scaled="keypoints_scaled",
                                  heatmap_keypoints="heatmap_keypoints", keypoints_in_image="keypoints_in_image", bbox_mask="bbox_mask",
                                  out_heatmaps="out_heatmaps", out_kpt_weights="out_kpt_weights")

    def __init__(self, meta_keys: Sequence[str] = ("id", "img_id", "img_path", "category_id", "crowd_index", "ori_shape", "img_shape",
                                                   "input_size", "input_center", "input_scale", "flip", "flip_direction", "flip_indices",
                                                   "raw_ann_info", "dataset_name"), pack_transformed: bool = False):
        self.meta_keys, self.pack_transformed = tuple(meta_keys), pack_transformed

    def transform(self, results: dict) -> dict:
        from .structures import InstanceData, PoseDataSample

        img = results["img"]
        if isinstance(img, np.ndarray):  # image_to_tensor: HWC -> CHW
            img = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1) if img.ndim == 3 else img[None]))
        ds = PoseDataSample()
        gt = InstanceData()
        for key, packed in results.get("instance_mapping_table", self.instance_mapping_table).items():
            if key in results:
                gt.set_field(results[key], packed)
        if self.pack_transformed and "transformed_keypoints" in results:
            gt.set_field(results["transformed_keypoints"], "transformed_keypoints")
        ds.gt_instances = gt
        ds.set_metainfo({k: results[k] for k in self.meta_keys if k in results})
        return dict(inputs=img, data_samples=ds)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(meta_keys={self.meta_keys}, pack_transformed={self.pack_transformed})"


class Compose:
    """mmengine.dataset.Compose [3P]: a list of transforms (config dicts built through ``TRANSFORMS``, or callables) applied in
    order; a transform returning ``None`` ends the chain. ``batched`` applies the chain to a list of samples and lets
    ``TopdownAffine`` warp all boxes that share a source image in one launch."""

    def __init__(self, transforms):
        self.transforms = []
        for t in transforms or []:
            if isinstance(t, dict):
                t = TRANSFORMS.build(t)
            elif not callable(t):
                raise TypeError(f"transform should be a callable object or dict, but got {type(t)}")
            self.transforms.append(t)

    def __call__(self, data: dict) -> Optional[dict]:
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def batched(self, data_list: List[dict]) -> List[dict]:
        for t in self.transforms:
            if isinstance(t, TopdownAffine):
                data_list = t.transform_batch(data_list)
            else:
                data_list = [d for d in (t(x) for x in data_list) if d is not None]
        return data_list

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join(f"\n    {t}" for t in self.transforms) + "\n)"


def pseudo_collate(data_list: List[dict]) -> dict:
    """mmengine.dataset.pseudo_collate [3P] for the packed samples: lists, no stacking."""
    return dict(inputs=[d["inputs"] for d in data_list], data_samples=[d["data_samples"] for d in data_list])
