"""Golden vectors for the crop-pipeline box arithmetic from the REFERENCE's own functions (build container only):
``bbox_xyxy2cs``, ``bbox_xywh2xyxy``, ``get_udp_warp_matrix`` (mmpose/structures/bbox/transforms.py). The image warp
itself (cv2.warpAffine) cannot be pinned here: cv2 is not installed."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference_bbox  # noqa: E402


def main():
    t = load_reference_bbox()
    rng = np.random.default_rng(20250931)
    n = 24
    xy0 = rng.uniform(-20, 500, (n, 2))
    wh = rng.uniform(5, 400, (n, 2))
    boxes = np.concatenate([xy0, xy0 + wh], -1).astype(np.float32)
    boxes[0] = [0, 0, 640, 480]
    out = dict(boxes_xyxy=boxes, boxes_xywh=np.concatenate([xy0, wh], -1).astype(np.float32))
    out["xywh2xyxy"] = t.bbox_xywh2xyxy(out["boxes_xywh"].copy())
    for pad in (1.0, 1.25):
        c, s = t.bbox_xyxy2cs(boxes, padding=pad)
        out[f"center_p{pad}"], out[f"scale_p{pad}"] = c, s
    mats = []
    c, s = t.bbox_xyxy2cs(boxes, padding=1.25)
    for i in range(n):
        w, h = s[i]
        sc = np.array([w, w / 0.75], np.float32) if w > h * 0.75 else np.array([h * 0.75, h], np.float32)
        rot = 0.0 if i % 3 else float(rng.uniform(-40, 40))
        mats.append(t.get_udp_warp_matrix(c[i], sc, rot, output_size=(192, 256)))
        out.setdefault("rot", []).append(rot)
        out.setdefault("fixed_scale", []).append(sc)
    out["rot"] = np.array(out["rot"])
    out["fixed_scale"] = np.stack(out["fixed_scale"])
    out["udp_mats"] = np.stack(mats)
    np.savez_compressed(os.path.join(HERE, "warp_boxes.npz"), **out)
    print("warp_boxes.npz", os.path.getsize(os.path.join(HERE, "warp_boxes.npz")))


if __name__ == "__main__":
    main()
