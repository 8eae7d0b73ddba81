"""Generate the committed golden fixtures from the REFERENCE's own functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every expected output below is produced by code loaded from the reference tree
(``_ref_import.load_reference``): ``get_heatmap_expected_value`` / ``_prepare_oks_kernels``
(mmpose/codecs/utils/post_processing.py), ``ProbMap.decode`` (mmpose/codecs/probmap.py) and
``flip_heatmaps`` (mmpose/models/utils/tta.py). Inputs are synthetic and seeded. The
fixtures are data only (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402

K = 17


def blobs(rng, H, W, n_blobs=(1, 3), sigma=(0.6, 2.5), support_cut=0.02, offgrid=True):
    """Sparsemax-like maps: a few truncated Gaussian bumps, exact zeros elsewhere, sum 1."""
    out = np.zeros((K, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for k in range(K):
        m = np.zeros((H, W))
        for _ in range(rng.integers(n_blobs[0], n_blobs[1] + 1)):
            cx = rng.uniform(-2, W + 1) if offgrid else rng.uniform(3, W - 4)
            cy = rng.uniform(-2, H + 1) if offgrid else rng.uniform(3, H - 4)
            s = rng.uniform(*sigma)
            m += rng.uniform(0.3, 1.0) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        m[m < support_cut * m.max()] = 0
        if m.sum() > 0:
            m /= m.sum()
        out[k] = m.astype(np.float32)
    return out


def border_peaks(rng, H, W):
    """Single hot pixels / tiny bumps on rows/cols 0, 1, H-2, H-1 and the four corners."""
    out = np.zeros((K, H, W), np.float32)
    spots = [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, W // 2), (H - 1, W // 3), (H // 2, 0),
             (H // 3, W - 1), (1, 1), (H - 2, W - 2), (1, W // 2), (H // 2, 1), (H - 2, 5), (7, W - 2),
             (H // 2, W // 2), (0, 1), (H - 1, W - 2)]
    for k, (y, x) in enumerate(spots):
        out[k, y, x] = rng.uniform(0.2, 0.9)
        # a weaker neighbour so that the smoothed peak is asymmetric
        y2 = min(max(y + rng.integers(-1, 2), 0), H - 1)
        x2 = min(max(x + rng.integers(-1, 2), 0), W - 1)
        out[k, y2, x2] += rng.uniform(0.05, 0.2)
    return out


def plateaus(rng, H, W):
    """All-zero maps, constant maps, 2x2 and row plateaus: exercises argmax tie-breaks and
    the zero-second-derivative substitution."""
    out = np.zeros((K, H, W), np.float32)
    out[1] = 1.0 / (H * W)
    out[2] = 0.25
    out[3, 10:12, 20:22] = 0.25
    out[4, 30, :] = 1.0 / W
    out[5, :, 7] = 1.0 / H
    out[6, 5, 5] = 0.5
    out[6, 50, 40] = 0.5
    out[7, 20, 10] = 0.5
    out[7, 20, 12] = 0.5
    out[8, H // 2, W // 2] = 1.0
    out[9, 0:3, 0:3] = 1.0 / 9
    out[10, H - 3 :, W - 3 :] = 1.0 / 9
    out[11] = rng.integers(0, 2, (H, W)).astype(np.float32) / 4
    out[12, 31:33, 23:25] = 0.25
    # 13..16 stay zero
    return out


def cropcoco_maps(ns, rng, H, W, sharpen):
    """CropCOCO-style targets (BASELINE config 5): the reference's own `generate_probmaps`
    (mmpose/codecs/utils/oks_map.py:9-67) for ground-truth keypoints of which ~25 % lie OUTSIDE the
    heatmap (crop cuts the person), then sharpened + renormalised to look like a Sparsemax output."""
    kpts = np.stack([rng.uniform(-0.3 * W, 1.3 * W, K), rng.uniform(-0.3 * H, 1.3 * H, K)], -1)[None]
    kpts[0, :4] = [[W / 2, H / 2], [0.0, 0.0], [W - 1.0, H - 1.0], [-6.0, H / 3]]
    hm, _ = ns.utils.generate_probmaps((W, H), kpts, np.ones((1, K), np.float32), sigma=-1)
    hm = np.maximum(hm - sharpen, 0).astype(np.float32)
    s = hm.reshape(K, -1).sum(-1)[:, None, None]
    return (hm / np.where(s > 0, s, 1)).astype(np.float32)


def run_case(ns, codec, hm):
    locs, vals, conv = ns.post.get_heatmap_expected_value(hm.copy(), return_heatmap=True)
    kpts, scores = codec.decode(hm)
    return dict(hm=hm, locs=locs, vals=vals, conv=conv, keypoints=kpts, scores=scores)


def main():
    ns = load_reference()
    rng = np.random.default_rng(20250929)
    codec_s = ns.KEYPOINT_CODECS.build(dict(type="ProbMap", input_size=(192, 256), heatmap_size=(48, 64), sigma=-1))
    codec_b = ns.KEYPOINT_CODECS.build(dict(type="ProbMap", input_size=(288, 384), heatmap_size=(72, 96), sigma=-1))
    assert codec_s.support_batch_decoding is False

    cases = {
        "s_blobs0": (codec_s, blobs(rng, 64, 48)),
        "s_blobs1": (codec_s, blobs(rng, 64, 48, n_blobs=(2, 4), sigma=(0.4, 1.2))),
        "s_blobs_interior": (codec_s, blobs(rng, 64, 48, offgrid=False)),
        "s_border": (codec_s, border_peaks(rng, 64, 48)),
        "s_plateau": (codec_s, plateaus(rng, 64, 48)),
        "s_noise": (codec_s, rng.random((K, 64, 48), dtype=np.float32)),
        "s_cropcoco0": (codec_s, cropcoco_maps(ns, rng, 64, 48, 0.5)),
        "s_cropcoco1": (codec_s, cropcoco_maps(ns, rng, 64, 48, 0.05)),
        "b_blobs0": (codec_b, blobs(rng, 96, 72, sigma=(0.8, 3.5))),
        "b_border": (codec_b, border_peaks(rng, 96, 72)),
    }
    out = {}
    for name, (codec, hm) in cases.items():
        for key, val in run_case(ns, codec, hm).items():
            out[f"{name}/{key}"] = val
    np.savez_compressed(os.path.join(HERE, "decode_cases.npz"), **out)

    kern = {}
    for tag, (H, W) in {"s": (64, 48), "b": (96, 72)}.items():
        for k, w in enumerate(ns.post._prepare_oks_kernels(K, H, W)):
            kern[f"{tag}/{k}"] = w[0]
    np.savez_compressed(os.path.join(HERE, "oks_kernels.npz"), **kern)

    # flip_heatmaps (tta.py:35-39) with the COCO flip pairs, shift_heatmap=False as in the config
    flip_indices = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    x = torch.from_numpy(rng.random((2, K, 8, 6), dtype=np.float32))
    y = ns.tta.flip_heatmaps(x.clone(), flip_mode="heatmap", flip_indices=flip_indices, shift_heatmap=False)
    np.savez_compressed(
        os.path.join(HERE, "flip_heatmaps.npz"), x=x.numpy(), y=y.numpy(), flip_indices=np.array(flip_indices)
    )

    # batched (B>1) call is broken in the reference: record that it raises
    try:
        ns.post.get_heatmap_expected_value(np.zeros((2, K, 64, 48), np.float32))
        batched = "ok"
    except Exception as e:  # noqa: BLE001
        batched = type(e).__name__
    np.savez_compressed(os.path.join(HERE, "quirks.npz"), batched_call_raises=np.array(batched))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
