"""The val-pipeline transforms in front of the hot path, for ``apis.inference_topdown``:
``GetBBoxCenterScale`` -> ``TopdownAffine`` (UDP) -> ``PackPoseInputs``. The box arithmetic is host numpy exactly as in
the reference (it is a handful of scalars per person); the image warp runs on the device (``pp_warp_affine_u8``).
"""
import math
from typing import Tuple

import numpy as np
import torch

from . import _lib


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
