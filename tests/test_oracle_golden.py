"""CPU: the oracle restatement of the decode vs the fixtures generated from the reference's own
functions (tests/golden/make_golden.py). Bit-exact -- this is what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import decode_ref as D

CASES = ["s_blobs0", "s_blobs1", "s_blobs_interior", "s_border", "s_plateau", "s_noise", "s_cropcoco0", "s_cropcoco1",
         "b_blobs0", "b_border"]


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "decode_cases.npz"))


def _sizes(name):
    return ((192, 256), (48, 64)) if name.startswith("s_") else ((288, 384), (72, 96))


@pytest.mark.parametrize("tag,hw", [("s", (64, 48)), ("b", (96, 72))])
def test_oks_kernels_bit_exact(golden_dir, tag, hw):
    ref = np.load(os.path.join(golden_dir, "oks_kernels.npz"))
    mine = D.oks_kernels(17, *hw)
    for k in range(17):
        assert mine[k].dtype == np.float64
        assert np.array_equal(mine[k], ref[f"{tag}/{k}"]), f"kernel {k}"
    # diameters quoted in SURVEY 8a11
    want = [5] * 5 + [15, 15, 13, 13, 11, 11] + [19] * 6 if tag == "s" else [5] * 5 + [19] * 4 + [15, 15] + [19] * 6
    assert [m.shape[0] for m in mine] == want


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("backend", ["symmetric_f64", "scipy"])
def test_expected_value_bit_exact(cases, name, backend):
    hm = cases[f"{name}/hm"]
    locs, vals, conv = D.heatmap_expected_value(hm, backend=backend, return_conv=True)
    assert conv.dtype == np.float32 and locs.dtype == np.float32 and vals.dtype == np.float32
    assert np.array_equal(conv, cases[f"{name}/conv"])
    assert np.array_equal(locs, cases[f"{name}/locs"], equal_nan=True)
    assert np.array_equal(vals, cases[f"{name}/vals"])


@pytest.mark.parametrize("name", CASES)
def test_codec_decode_bit_exact(cases, name):
    input_size, heatmap_size = _sizes(name)
    kpts, scores = D.probmap_decode(cases[f"{name}/hm"], input_size, heatmap_size)
    assert kpts.dtype == np.float64 and kpts.shape == (1, 17, 2)
    assert scores.dtype == np.float32 and scores.shape == (1, 17)
    assert np.array_equal(kpts, cases[f"{name}/keypoints"], equal_nan=True)
    assert np.array_equal(scores, cases[f"{name}/scores"])


def test_flip_back_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "flip_heatmaps.npz"))
    assert tuple(g["flip_indices"]) == D.COCO_FLIP_INDICES
    assert np.array_equal(D.flip_back(g["x"], g["flip_indices"]), g["y"])


def test_flip_back_with_shift_matches_reference(golden_dir):
    # flip_heatmaps(..., shift_heatmap=True) of the reference (tta.py:64-66; tests/golden/make_golden_flip_shift.py)
    g = np.load(os.path.join(golden_dir, "flip_heatmaps_shift.npz"))
    assert np.array_equal(D.flip_back(g["x"], g["flip_indices"], shift_heatmap=True), g["y_shift"])
    assert np.array_equal(D.flip_back(g["x"], g["flip_indices"]), g["y_plain"])


def test_reference_batched_call_is_broken(golden_dir):
    # the reference's own B>1 form raises; the oracle therefore only restates the 3-D form
    assert str(np.load(os.path.join(golden_dir, "quirks.npz"))["batched_call_raises"]) == "ValueError"
    with pytest.raises(AssertionError):
        D.heatmap_expected_value(np.zeros((2, 17, 64, 48), np.float32))


def test_symmetric_padding_is_scipy_reflect():
    # scipy 'reflect' == numpy 'symmetric' != torch/numpy 'reflect'
    rng = np.random.default_rng(3)
    hm = rng.random((20, 24), dtype=np.float32)
    k = D.oks_kernels(17, 64, 48)[11]  # 19x19
    assert np.array_equal(D.convolve_symmetric_f64(hm, k), D.convolve_scipy(hm, k))
