#!/usr/bin/env python
"""Differential fuzz of the HIP crop warp (pp_warp.hip through transforms.warp_affine_crops) against oracle/warp_ref.py (the restatement of
cv2.warpAffine's fixed-point bilinear path): random image sizes (down to 1 x 1 .. 2 x 2), boxes inside / across / wholly outside the image, slivers,
huge boxes, rotations, both input sizes - crops equal byte for byte.   python tests/fuzz_warp.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import warp_ref  # noqa: E402
from probpose_code_amd import transforms as T  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n, bad, seed = 0, 0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    rng = np.random.default_rng(9000 + seed)
    seed += 1
    h, w = [(480, 640), (97, 131), (7, 5), (2, 2), (1, 3), (720, 1280), (33, 1000)][int(rng.integers(0, 7))]
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out_wh = [(192, 256), (288, 384)][int(rng.integers(0, 2))]
    nb = int(rng.integers(1, 7))
    boxes = np.zeros((nb, 4), np.float32)
    for i in range(nb):
        kind = rng.random()
        if kind < 0.5:
            x0, y0 = rng.uniform(0, w), rng.uniform(0, h)
            bw, bh = rng.uniform(1, w), rng.uniform(1, h)
        elif kind < 0.7:
            x0, y0, bw, bh = rng.uniform(-2 * w, 2 * w), rng.uniform(-2 * h, 2 * h), rng.uniform(1, 3 * w), rng.uniform(1, 3 * h)
        elif kind < 0.85:
            x0, y0, bw, bh = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0.01, 1.0), rng.uniform(1, h)
        else:
            x0, y0, bw, bh = rng.uniform(-w, 0), rng.uniform(-h, 0), 10 * w, 10 * h
        boxes[i] = [x0, y0, x0 + bw, y0 + bh]
    c, s, mats = T.topdown_affine_params(boxes, out_wh)
    for i in range(nb):
        if rng.random() < 0.3:
            mats[i] = T.get_udp_warp_matrix(c[i], s[i], float(rng.uniform(-180, 180)), out_wh)
    crops = T.warp_affine_crops(torch.from_numpy(img).cuda(), mats, out_wh).cpu().numpy()
    for i in range(nb):
        ref = warp_ref.warp_affine_u8(img, mats[i], out_wh).transpose(2, 0, 1)
        n += 1
        if not np.array_equal(crops[i], ref):
            bad += 1
            d = np.abs(crops[i].astype(int) - ref.astype(int))
            print(f"MISMATCH seed {seed - 1} box {i} image {h}x{w}: {int((d > 0).sum())} bytes differ, max {int(d.max())}", flush=True)
print(f"{n} crops over {seed} images in {seconds:.0f} s, {bad} mismatches")
print("WARP FUZZ", "FAILED" if bad else "OK")
