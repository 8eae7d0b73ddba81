"""Generates tests/golden/flip_heatmaps_shift.npz from the REFERENCE's own ``flip_heatmaps`` (mmpose/models/utils/tta.py:9-67,
imported by file path: torch is its only dependency) with ``shift_heatmap=True`` (:64-66) - the case the ProbPose config does
not use and round 4 built into the fused flip-merge + decode kernel. Run in the build container (the reference does not travel):
    python tests/golden/make_golden_flip_shift.py
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PROBPOSE_REFERENCE", "/root/reference")

spec = importlib.util.spec_from_file_location("ref_tta", os.path.join(REF, "mmpose", "models", "utils", "tta.py"))
tta = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tta)

rng = np.random.default_rng(20260930)
flip_indices = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
x = torch.from_numpy(rng.random((3, 17, 8, 12), dtype=np.float32))
y_shift = tta.flip_heatmaps(x.clone(), flip_mode="heatmap", flip_indices=flip_indices, shift_heatmap=True)
y_plain = tta.flip_heatmaps(x.clone(), flip_mode="heatmap", flip_indices=flip_indices, shift_heatmap=False)
np.savez_compressed(os.path.join(HERE, "flip_heatmaps_shift.npz"), x=x.numpy(), y_shift=y_shift.numpy(), y_plain=y_plain.numpy(),
                    flip_indices=np.array(flip_indices))
print("wrote flip_heatmaps_shift.npz", tuple(y_shift.shape))
