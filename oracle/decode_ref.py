"""Oracle (test infrastructure, NOT product): CPU restatement of the ProbMap decode.

Follows, step by step, the arithmetic of the reference (paths relative to the reference
tree):

* OKS kernels ............ ``mmpose/codecs/utils/post_processing.py:13-39``
* expected-OKS decode .... ``mmpose/codecs/utils/post_processing.py:308-381``
* sub-pixel Newton step .. ``mmpose/codecs/utils/post_processing.py:384-430``
* codec rescale .......... ``mmpose/codecs/probmap.py:170-220`` (gaussian branch, rescale ``:218``)
* flip-back .............. ``mmpose/models/utils/tta.py:35-39`` (``flip_mode='heatmap'``, no shift)
* image-space mapping .... ``mmpose/models/pose_estimators/topdown.py:165-167``

Two interchangeable convolution back-ends are provided so that each validates the other:

* ``convolve_symmetric_f64`` -- hand-rolled: half-sample-symmetric padding (what
  ``scipy.ndimage`` calls ``mode='reflect'``), float64 accumulation in raster order over
  the kernel taps, one rounding to float32 at the end;
* ``convolve_scipy`` -- the very call the reference makes (``scipy.ndimage.convolve``,
  a third-party dependency present in this image).

Pinned: ``tests/test_oracle_golden.py`` checks every function here bit-for-bit against the
fixtures that ``tests/golden/make_golden.py`` produced from the reference's own functions.
"""
from typing import List, Sequence, Tuple

import numpy as np

# COCO per-keypoint sigmas x100 as hard-wired in post_processing.py:16
_COCO_SIGMAS_X100 = (2.6, 2.5, 2.5, 3.5, 3.5, 7.9, 7.9, 7.2, 7.2, 6.2, 6.2, 10.7, 10.7, 8.7, 8.7, 8.9, 8.9)

COCO_FLIP_INDICES = (0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15)


def oks_kernels(K: int, H: int, W: int) -> List[np.ndarray]:
    """Normalised Gaussian "OKS kernels", one (d, d) float64 array per keypoint.

    post_processing.py:13-39. The operation order (sqrt, then square again, exp, one
    division by the 2-D sum) is kept so that the weights are bit-identical.
    """
    area = np.sqrt(H / 1.25 * W / 1.25)
    sig = np.array(_COCO_SIGMAS_X100) / 100
    out = []
    for k in range(K):
        s = (sig[k] * 2) ** 2 * area * 2
        s = np.clip(s, 0.55, 3.0)
        r = int(np.ceil(s * 3))
        d = 2 * r + 1
        ax = np.arange(d) - d // 2
        gx, gy = np.meshgrid(ax, ax)
        dist = np.sqrt(gx**2 + gy**2)
        w = np.exp(-(dist**2) / (2 * s))
        out.append(w / w.sum())
    return out


def convolve_symmetric_f64(hm: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """2-D convolution of a float32 (H, W) map with a centred odd (d, d) float64 kernel.

    Boundary: half-sample symmetric (``d c b a | a b c d | d c b a``) == numpy
    ``mode='symmetric'`` == scipy.ndimage ``mode='reflect'``. Accumulates in
    float64 over the taps in raster order -- the order ``scipy.ndimage`` walks its
    footprint -- and rounds to float32 once. The kernels are point-symmetric, so
    convolution == correlation.
    """
    assert hm.ndim == 2 and kern.ndim == 2 and kern.shape[0] == kern.shape[1] and kern.shape[0] % 2 == 1
    H, W = hm.shape
    d = kern.shape[0]
    r = d // 2
    pad = np.pad(hm.astype(np.float64), r, mode="symmetric")
    acc = np.zeros((H, W), np.float64)
    for i in range(d):
        for j in range(d):
            acc += pad[i : i + H, j : j + W] * kern[i, j]
    return acc.astype(np.float32)


def convolve_scipy(hm: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """The reference's own call (post_processing.py:351) on one (H, W) map."""
    from scipy.ndimage import convolve

    return convolve(hm[None], kern[None], mode="reflect")[0]


def subpixel_refine(conv: np.ndarray, locs: np.ndarray) -> np.ndarray:
    """One Newton step per axis on the convolved maps (post_processing.py:384-430).

    conv: (N, H, W) float32, locs: (N, 2) float32 integer-valued (x, y). Interior peaks
    only (0 < x < W-1 and 0 < y < H-1); zero second derivatives are replaced by 1e-6;
    the shift is NOT clamped. All arithmetic in float32, as numpy does it there.
    """
    N, H, W = conv.shape
    out = locs.copy()
    xs = locs[:, 0].astype(np.int32)
    ys = locs[:, 1].astype(np.int32)
    for n in range(N):
        x, y = int(xs[n]), int(ys[n])
        if not (0 < x < W - 1 and 0 < y < H - 1):
            continue
        h = conv[n]
        two = np.float32(2.0)
        dx = (h[y, x + 1] - h[y, x - 1]) / two
        dy = (h[y + 1, x] - h[y - 1, x]) / two
        dxx = h[y, x + 1] + h[y, x - 1] - two * h[y, x]
        dyy = h[y + 1, x] + h[y - 1, x] - two * h[y, x]
        if dxx == 0:
            dxx = np.float32(1e-6)
        if dyy == 0:
            dyy = np.float32(1e-6)
        out[n, 0] = out[n, 0] + (-dx / dxx)
        out[n, 1] = out[n, 1] + (-dy / dyy)
    return out


def heatmap_expected_value(
    heatmaps: np.ndarray, backend: str = "symmetric_f64", return_conv: bool = False
) -> Tuple[np.ndarray, ...]:
    """(K, H, W) float32 -> locs (K, 2) float32 [x, y], vals (K,) float32.

    post_processing.py:308-381 for the 3-D input case (the only one that works there,
    SURVEY H8): per-keypoint OKS-kernel convolution, flat first-occurrence argmax of the
    convolved map, sub-pixel step; ``vals`` is the RAW (un-convolved) map at the integer
    argmax.
    """
    assert isinstance(heatmaps, np.ndarray) and heatmaps.ndim == 3, "expects (K, H, W)"
    K, H, W = heatmaps.shape
    kernels = oks_kernels(K, H, W)
    conv_fn = {"symmetric_f64": convolve_symmetric_f64, "scipy": convolve_scipy}[backend]
    conv = np.zeros_like(heatmaps)
    for k in range(K):
        conv[k] = conv_fn(heatmaps[k], kernels[k])
    flat = np.argmax(conv.reshape(K, H * W), axis=1)
    ys, xs = np.unravel_index(flat, (H, W))
    locs = np.stack((xs, ys), axis=-1).astype(np.float32)
    locs = subpixel_refine(conv, locs)
    vals = heatmaps[np.arange(K), ys, xs]
    if return_conv:
        return locs, vals, conv
    return locs, vals


def probmap_decode(
    heatmaps: np.ndarray,
    input_size: Sequence[int] = (192, 256),
    heatmap_size: Sequence[int] = (48, 64),
    backend: str = "symmetric_f64",
) -> Tuple[np.ndarray, np.ndarray]:
    """``ProbMap.decode`` (probmap.py:170-220): keypoints (1, K, 2) float64 in input-pixel
    space, scores (1, K) float32. Note the reference's rescale ``/(W-1, H-1) * input_size``
    (probmap.py:218; quirk H8) -- float32 locs divided by an int list promote to float64."""
    W, H = heatmap_size
    locs, vals = heatmap_expected_value(np.array(heatmaps, copy=True), backend=backend)
    kpts = locs[None] / [W - 1, H - 1] * input_size
    return kpts, vals[None]


def flip_back(heatmaps: np.ndarray, flip_indices: Sequence[int] = COCO_FLIP_INDICES, shift_heatmap: bool = False) -> np.ndarray:
    """tta.py:35-39: mirror the last axis, then permute the keypoint channels (B, K, H, W); ``shift_heatmap`` (:64-66):
    ``heatmaps[..., 1:] = heatmaps[..., :-1].clone()`` - one pixel to the right, column 0 keeps its value."""
    out = heatmaps[..., ::-1][:, list(flip_indices)]
    if shift_heatmap:
        out = np.concatenate([out[..., :1], out[..., :-1]], axis=-1)
    return out


def tta_average(hm: np.ndarray, hm_flipped_pass: np.ndarray, flip_indices=COCO_FLIP_INDICES, shift_heatmap: bool = False) -> np.ndarray:
    """probmap_head.py:757-763: ``(htm + flip_heatmaps(htm_flip)) * 0.5`` in float32."""
    return ((hm + flip_back(hm_flipped_pass, flip_indices, shift_heatmap)) * np.float32(0.5)).astype(np.float32)


def to_image_space(kpts, input_size, input_center, input_scale):
    """topdown.py:165-167: input-pixel space -> image space."""
    return kpts / input_size * input_scale + input_center - 0.5 * input_scale
