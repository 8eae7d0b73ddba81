"""``ProbMap`` keypoint codec on MI355X (reference: ``mmpose/codecs/probmap.py:19-220``,
``mmpose/codecs/base.py:9-77``).

Same constructor arguments, same ``decode`` contract (``(K, H, W)`` float32 numpy in,
``(1, K, 2)`` float64 keypoints in input-pixel space and ``(1, K)`` float32 scores out) --
but the arithmetic (OKS-kernel convolution, argmax, sub-pixel step, rescale) runs in the
fused HIP kernel ``pp_probmap_decode`` for a whole batch at once, optionally together with
the flip-test average. ``batch_decode`` is overridden, so ``support_batch_decoding`` is
True and ``BaseHead.decode`` (``mmpose/models/heads/base_head.py:57-62``) takes the batched
branch instead of the per-sample Python loop.
"""
from abc import ABCMeta, abstractmethod
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .registry import KEYPOINT_CODECS, register

# COCO keypoint sigmas (x100) that the reference hard-wires into its OKS kernels
# (mmpose/codecs/utils/post_processing.py:16); the decode is therefore limited to K <= 17.
_KPT_SIGMAS = np.array([2.6, 2.5, 2.5, 3.5, 3.5, 7.9, 7.9, 7.2, 7.2, 6.2, 6.2, 10.7, 10.7, 8.7, 8.7, 8.9, 8.9]) / 100


def oks_kernel_taps(K: int, H: int, W: int) -> Tuple[np.ndarray, np.ndarray]:
    """Separable factors of the reference's OKS kernels (post_processing.py:13-39).

    The reference builds, per keypoint, ``exp(-d^2 / 2s) / sum`` on a (2r+1)^2 grid with
    ``s = clip((2 sigma_k)^2 * sqrt(H/1.25 * W/1.25) * 2, 0.55, 3.0)`` and ``r = ceil(3 s)``.
    That kernel is the outer product of ``g / sum(g)``, ``g[t] = exp(-t^2 / 2s)``, with
    itself; the HIP kernel convolves rows then columns with this vector in float64, which
    rounds to the same float32 map (tests/test_decode_parity.py checks the maps bit for bit).

    Returns ``taps`` (K, PP_MAX_TAPS) float64 (row k: 2 r_k + 1 factors, zero padded) and
    ``radius`` (K,) int32.
    """
    if K > len(_KPT_SIGMAS):
        raise IndexError(f"ProbMap decode is defined for at most {len(_KPT_SIGMAS)} (COCO) keypoints, got {K}")
    area = np.sqrt(H / 1.25 * W / 1.25)
    taps = np.zeros((K, _lib.PP_MAX_TAPS), np.float64)
    radius = np.zeros((K,), np.int32)
    for k in range(K):
        s = float(np.clip((_KPT_SIGMAS[k] * 2) ** 2 * area * 2, 0.55, 3.0))
        r = int(np.ceil(s * 3))
        t = np.arange(-r, r + 1, dtype=np.float64)
        g = np.exp(-(t**2) / (2 * s))
        taps[k, : 2 * r + 1] = g / g.sum()
        radius[k] = r
    return taps, radius


class BaseKeypointCodec(metaclass=ABCMeta):
    """base.py:9-77."""

    auxiliary_encode_keys = set()
    field_mapping_table: Dict[str, str] = dict()
    instance_mapping_table: Dict[str, str] = dict()
    label_mapping_table: Dict[str, str] = dict()

    @abstractmethod
    def encode(self, keypoints: np.ndarray, keypoints_visible: Optional[np.ndarray] = None) -> dict:
        """Encode keypoints."""

    @abstractmethod
    def decode(self, encoded: Any) -> Tuple[np.ndarray, np.ndarray]:
        """Decode keypoints."""

    def batch_decode(self, batch_encoded: Any) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        raise NotImplementedError()

    @property
    def support_batch_decoding(self) -> bool:
        return getattr(type(self), "batch_decode") is not BaseKeypointCodec.batch_decode


@register(KEYPOINT_CODECS, reference_name="ProbMap", mi355x_name="ProbMapMI355X")
class ProbMap(BaseKeypointCodec):
    """Expected-OKS probability-map codec; decode on the GPU.

    Args are those of the reference (probmap.py:71-96): ``input_size`` [w, h],
    ``heatmap_size`` [W, H], ``heatmap_type`` in {"gaussian", "combined"} (only
    "gaussian" -- the one ProbPose uses -- is decodable here), ``sigma``,
    ``radius_factor``, ``blur_kernel_size``, ``increase_sigma_with_padding``.
    """

    label_mapping_table = dict(keypoint_weights="keypoint_weights")
    field_mapping_table = dict(heatmaps="heatmaps")

    def __init__(
        self,
        input_size: Tuple[int, int],
        heatmap_size: Tuple[int, int],
        heatmap_type: str = "gaussian",
        sigma: float = 2.0,
        radius_factor: float = 0.0546875,
        blur_kernel_size: int = 11,
        increase_sigma_with_padding=False,
    ) -> None:
        super().__init__()
        self.input_size = input_size
        self.heatmap_size = heatmap_size
        self.radius_factor = radius_factor
        self.heatmap_type = heatmap_type
        self.blur_kernel_size = blur_kernel_size
        self.scale_factor = ((np.array(input_size) - 1) / (np.array(heatmap_size) - 1)).astype(np.float32)
        self.increase_sigma_with_padding = increase_sigma_with_padding
        self.sigma = sigma
        if self.heatmap_type not in {"gaussian", "combined"}:
            raise ValueError(
                f"{self.__class__.__name__} got invalid `heatmap_type` value"
                f"{self.heatmap_type}. Should be one of "
                '{"gaussian", "combined"}'
            )
        self._tables: Dict[Tuple[int, int, int, str], Tuple[torch.Tensor, torch.Tensor]] = {}

    # -- encode is the training-side label generator (GenerateTarget); not on the inference path
    def encode(self, keypoints, keypoints_visible=None, id_similarity=0.0, keypoints_visibility=None) -> dict:
        raise NotImplementedError(
            "ProbMap.encode (probmap.py:98-168) generates training targets and is outside the "
            "MI355X inference hot path (SURVEY.md 2: 'decode only; encode is training')."
        )

    # -- device-side tables, built once per (K, H, W, device) instead of on every call (post_processing.py:344)
    def _device_tables(self, K: int, H: int, W: int, device: torch.device):
        key = (K, H, W, str(device))
        if key not in self._tables:
            taps, radius = oks_kernel_taps(K, H, W)
            self._tables[key] = (torch.from_numpy(taps).to(device), torch.from_numpy(radius).to(device))
        return self._tables[key]

    def decode_device(
        self,
        heatmaps: torch.Tensor,
        heatmaps_flip: Optional[torch.Tensor] = None,
        flip_indices: Optional[Sequence[int]] = None,
        return_avg: bool = False,
        return_conv: bool = False,
        shift_heatmap: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Batched decode on device tensors; nothing is copied to the host.

        heatmaps (B, K, H, W) float32 on a CUDA/HIP device; ``heatmaps_flip`` (same shape)
        is the output of the horizontally flipped pass -- it is flipped back, channel-permuted
        by ``flip_indices`` and averaged inside the kernel (probmap_head.py:757-763); ``shift_heatmap``: the flipped-back
        map is moved one pixel to the right first (``flip_heatmaps(..., shift_heatmap=True)``, models/utils/tta.py:64-66).
        Returns device tensors: ``keypoints`` (B, K, 2) f64 input-pixel space, ``scores``
        (B, K) f32, ``locs`` (B, K, 2) f32 heatmap space and optionally ``heatmaps`` (the
        averaged maps) / ``conv`` (the OKS-convolved maps).
        """
        assert isinstance(heatmaps, torch.Tensor) and heatmaps.dim() == 4, "heatmaps should be a (B, K, H, W) tensor"
        if self.heatmap_type != "gaussian":
            raise NotImplementedError("only heatmap_type='gaussian' (the ProbPose setting) is decodable on MI355X")
        if not heatmaps.is_cuda:
            raise RuntimeError("ProbMap.decode_device needs tensors on the MI355X; there is no CPU fallback")
        B, K, H, W = heatmaps.shape
        Wc, Hc = self.heatmap_size
        assert (H, W) == (Hc, Wc), f"heatmap shape {(H, W)} does not match codec heatmap_size {(Hc, Wc)}"
        dev = heatmaps.device
        hm = heatmaps.contiguous().float()
        hmf = fi = None
        if heatmaps_flip is not None:
            assert heatmaps_flip.shape == heatmaps.shape
            assert flip_indices is not None and len(flip_indices) == K
            hmf = heatmaps_flip.contiguous().float()
            fi = self._flip_tensor(tuple(int(i) for i in flip_indices), dev)
        taps, radius = self._device_tables(K, H, W, dev)
        out = dict(
            locs=torch.empty((B, K, 2), dtype=torch.float32, device=dev),
            keypoints=torch.empty((B, K, 2), dtype=torch.float64, device=dev),
            scores=torch.empty((B, K), dtype=torch.float32, device=dev),
        )
        avg = torch.empty_like(hm) if return_avg else None
        conv = torch.empty_like(hm) if return_conv else None
        with torch.cuda.device(dev):
            _lib.call(
                "pp_probmap_decode_flags", _lib.ptr(hm), _lib.ptr(hmf), _lib.ptr(fi), _lib.ptr(taps), _lib.ptr(radius),
                B, K, H, W, float(self.input_size[0]), float(self.input_size[1]), 1.0, 1.0,
                _lib.ptr(avg), _lib.ptr(conv), _lib.ptr(out["locs"]), _lib.ptr(out["keypoints"]),
                _lib.ptr(out["scores"]), 4 if shift_heatmap else 0, _lib.stream_ptr(dev),
            )  # fmt: skip  (flags: PP_DECODE_SHIFT_HEATMAP = 4)
        if return_avg:
            out["heatmaps"] = avg
        if return_conv:
            out["conv"] = conv
        return out

    def _flip_tensor(self, flip_indices: Tuple[int, ...], device) -> torch.Tensor:
        key = ("flip", flip_indices, str(device))
        if key not in self._tables:
            self._tables[key] = torch.tensor(flip_indices, dtype=torch.int32, device=device)
        return self._tables[key]

    def batch_decode(self, batch_heatmaps: torch.Tensor) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """(B, K, H, W) device tensor -> per-sample lists, each element shaped as ``decode`` returns."""
        out = self.decode_device(batch_heatmaps)
        kpts = out["keypoints"].cpu().numpy()
        scores = out["scores"].cpu().numpy()
        return [kpts[i][None] for i in range(kpts.shape[0])], [scores[i][None] for i in range(scores.shape[0])]

    def decode(self, encoded: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """probmap.py:170-220 for one sample: (K, H, W) numpy -> ((1, K, 2) f64, (1, K) f32)."""
        if self.heatmap_type != "gaussian":
            raise NotImplementedError("only heatmap_type='gaussian' (the ProbPose setting) is decodable on MI355X")
        assert isinstance(encoded, np.ndarray), "heatmaps should be numpy.ndarray"
        assert encoded.ndim == 3, f"Invalid shape {encoded.shape}"
        hm = torch.from_numpy(np.ascontiguousarray(encoded, dtype=np.float32)).cuda()[None]
        kpts, scores = self.batch_decode(hm)
        return kpts[0], scores[0]
