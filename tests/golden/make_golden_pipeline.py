"""Golden vectors for the val pipeline's box / keypoint arithmetic from the REFERENCE's own ``TopdownAffine.transform``
(mmpose/datasets/transforms/topdown_transforms.py, build container only), loaded behind stubs for what this image lacks:

* ``cv2``: ``warpAffine`` records the matrix it is handed and returns zeros (the image warp is pinned elsewhere: the HIP
  kernel against the restated fixed-point arithmetic), ``transform`` / ``getAffineTransform`` are the affine map / the
  three-point solve in float64 as OpenCV computes them;
* ``mmcv.transforms.BaseTransform``, ``mmengine.is_seq_of``, a dict registry for ``mmpose.registry.TRANSFORMS``;
* ``mmpose.structures.bbox`` = the reference's own transforms.py.

GetBBoxCenterScale (common_transforms.py:57-85; its module imports half of mmcv) is applied here as its two statements on
the reference's ``bbox_xyxy2cs``. Output: tests/golden/val_pipeline_cases.npz."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def main():
    captured = []
    cv2 = R._shell("cv2")
    cv2.INTER_LINEAR = 1

    def warpAffine(img, M, size, flags=None):
        captured.append(np.array(M, copy=True))
        return np.zeros((size[1], size[0]) + img.shape[2:], img.dtype)

    def transform(pts, M):
        pts = np.asarray(pts)
        return (pts.astype(np.float64) @ np.asarray(M, np.float64)[:, :2].T + np.asarray(M, np.float64)[:, 2]).astype(pts.dtype)

    def getAffineTransform(src, dst):
        a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
        return np.linalg.solve(a, np.asarray(dst, np.float64)).T

    cv2.warpAffine, cv2.transform, cv2.getAffineTransform = warpAffine, transform, getAffineTransform
    bbox = R._load("_ref_bbox_transforms", "mmpose/structures/bbox/transforms.py")
    R._shell("mmpose", os.path.join(R.REF, "mmpose"))
    R._shell("mmpose.structures")
    sb = R._shell("mmpose.structures.bbox")
    for name in ("bbox_cs2xyxy", "bbox_xyxy2cs", "get_udp_warp_matrix", "get_warp_matrix"):
        setattr(sb, name, getattr(bbox, name))
    reg = R._shell("mmpose.registry")
    reg.TRANSFORMS = R._DictRegistry()
    mmcv = R._shell("mmcv")
    mt = R._shell("mmcv.transforms")

    class BaseTransform:
        def __call__(self, results):
            return self.transform(results)

    mt.BaseTransform = BaseTransform
    mmcv.transforms = mt
    mmengine = R._shell("mmengine")
    mmengine.is_seq_of = lambda seq, t: all(isinstance(v, t) for v in seq)
    td = R._load("_ref_topdown_transforms", "mmpose/datasets/transforms/topdown_transforms.py")

    rng = np.random.default_rng(20260929)
    n = 16
    xy0 = rng.uniform(-30, 400, (n, 2))
    wh = rng.uniform(8, 380, (n, 2))
    boxes = np.concatenate([xy0, xy0 + wh], -1).astype(np.float32)
    boxes[0] = [0, 0, 640, 480]
    kpts = rng.uniform(0, 480, (n, 1, 17, 2)).astype(np.float32)
    out = dict(boxes=boxes, keypoints=kpts, img_hw=np.array([480, 640]))
    for udp in (True, False):
        for pad_g, pad_i in ((1.25, 1.25), (1.0, 1.1)):
            t = td.TopdownAffine(input_size=(192, 256), input_padding=pad_i, use_udp=udp)
            keys = ("bbox_center", "bbox_scale", "input_center", "input_scale", "bbox_xyxy_wrt_input", "transformed_keypoints")
            acc = {k: [] for k in keys}
            mats = []
            for i in range(n):
                res = dict(img=np.zeros((480, 640, 3), np.uint8), bbox=boxes[i][None].copy(), bbox_score=np.ones(1, np.float32),
                           keypoints=kpts[i].copy())
                # GetBBoxCenterScale.transform (common_transforms.py:72-84)
                res["bbox_xyxy_wrt_input"] = res["bbox"]
                res["bbox_center"], res["bbox_scale"] = bbox.bbox_xyxy2cs(res["bbox"], padding=pad_g)
                captured.clear()
                res = t(res)
                mats.append(np.asarray(captured[0], np.float64))
                assert res["input_size"] == (192, 256) and res["img"].shape == (256, 192, 3) and res["bbox_mask"].shape == (1, 256, 192)
                for k in keys:
                    acc[k].append(np.asarray(res[k]))
            tag = f"udp{int(udp)}_g{pad_g}_i{pad_i}"
            for k in keys:
                out[f"{tag}/{k}"] = np.stack(acc[k])
            out[f"{tag}/warp_mat"] = np.stack(mats)
    path = os.path.join(HERE, "val_pipeline_cases.npz")
    np.savez_compressed(path, **out)
    print("val_pipeline_cases.npz", os.path.getsize(path), sorted(out)[:8])


if __name__ == "__main__":
    main()
