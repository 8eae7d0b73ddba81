"""CPU restatement of the crop extraction (TEST INFRASTRUCTURE): cv2.warpAffine(img, M, (w, h), flags=INTER_LINEAR) with
zero border as TopdownAffine calls it (mmpose/datasets/transforms/topdown_transforms.py:118-126).

PARITY UNPINNED: OpenCV is a third-party dependency of the reference (opencv-python, unpinned in requirements), absent
from this image, and the reference holds no golden image for the warp. The algorithm below is the published one
(opencv 4.x modules/imgproc/src/imgwarp.cpp: warpAffine -> WarpAffineInvoker -> remapBilinear<FixedPtCast<int, uchar, 15>>):
the inverse map in float64, source coordinates in fixed point with 5 fractional bits, 15-bit bilinear weights,
round-half-up at the end. The box arithmetic around it (matrices) IS pinned: tests/golden/warp_boxes.npz.
"""
import numpy as np


def invert_affine(m):
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


def _round(v):  # saturate_cast<int>(double) = cvRound: nearest, ties to even
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def warp_affine_u8(img, m, out_wh):
    """img (H, W, C) uint8, m (2, 3) forward matrix, out_wh = (w, h). Returns (h, w, C) uint8."""
    ih, iw, ic = img.shape
    w, h = out_wh
    M = invert_affine(m)
    xs, ys = np.arange(w), np.arange(h)
    adelta, bdelta = _round(M[0, 0] * xs * 1024.0), _round(M[1, 0] * xs * 1024.0)
    X0 = _round((M[0, 1] * ys + M[0, 2]) * 1024.0) + 16
    Y0 = _round((M[1, 1] * ys + M[1, 2]) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00, w01, w10, w11 = (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32
    pad = np.zeros((ih + 2, iw + 2, ic), np.int64)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):  # zero outside the image
        inside = (yy >= -1) & (yy <= ih) & (xx >= -1) & (xx <= iw)
        v = pad[np.clip(yy + 1, 0, ih + 1), np.clip(xx + 1, 0, iw + 1)]
        return np.where(inside[..., None], v, 0)

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None]
           + tap(sy + 1, sx + 1) * w11[..., None] + (1 << 14)) >> 15
    return np.clip(acc, 0, 255).astype(np.uint8)


# ---- heatmaps back on the image (mmpose/structures/utils.py:48-127, 146-175) -----------------------------------------
def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking three points onto three points, solved in float64 from float32 points
    (imgwarp.cpp; OpenCV is absent here - restated from the published algorithm, UNPINNED)."""
    src, dst = np.asarray(src, np.float32).astype(np.float64), np.asarray(dst, np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, 0:2], A[i, 2] = src[i], 1
        A[i + 3, 3:5], A[i + 3, 5] = src[i], 1
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def get_warp_matrix(center, scale, rot, output_size, inv=False):
    """mmpose/structures/bbox/transforms.py:362-426 with shift = 0 and fix_aspect_ratio=True: three point pairs (centre,
    a point half a width to the left rotated by ``rot``, and the perpendicular third point) -> getAffineTransform."""
    center, scale = np.asarray(center, np.float64), np.asarray(scale, np.float64)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    rad = np.deg2rad(rot)
    sn, cs = np.sin(rad), np.cos(rad)
    src_dir = np.array([[cs, -sn], [sn, cs]]) @ np.array([src_w * -0.5, 0.0])
    src, dst = np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32)
    src[0], src[1] = center, center + src_dir
    dst[0], dst[1] = [dst_w * 0.5, dst_h * 0.5], np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([dst_w * -0.5, 0.0])
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.r_[-d[1], d[0]]
    return get_affine_transform(dst, src) if inv else get_affine_transform(src, dst)


def warp_affine_f32(img, m, out_wh):
    """cv2.warpAffine for float32 images (h, w, C), INTER_LINEAR, zero border: the coordinate arithmetic of the uint8 path,
    float32 weights (1 - fy/32)(1 - fx/32), ... and a float32 sum of the four products."""
    ih, iw, ic = img.shape
    w, h = out_wh
    M = invert_affine(m)
    xs, ys = np.arange(w), np.arange(h)
    X = (_round((M[0, 1] * ys + M[0, 2]) * 1024.0)[:, None] + 16 + _round(M[0, 0] * xs * 1024.0)[None, :]) >> 5
    Y = (_round((M[1, 1] * ys + M[1, 2]) * 1024.0)[:, None] + 16 + _round(M[1, 0] * xs * 1024.0)[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = ((X & 31).astype(np.float32) / np.float32(32))[..., None], ((Y & 31).astype(np.float32) / np.float32(32))[..., None]
    one = np.float32(1)
    pad = np.zeros((ih + 2, iw + 2, ic), np.float32)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        inside = (yy >= -1) & (yy <= ih) & (xx >= -1) & (xx <= iw)
        return np.where(inside[..., None], pad[np.clip(yy + 1, 0, ih + 1), np.clip(xx + 1, 0, iw + 1)], np.float32(0))

    return (tap(sy, sx) * ((one - ay) * (one - ax)) + tap(sy, sx + 1) * ((one - ay) * ax) + tap(sy + 1, sx) * (ay * (one - ax))
            + tap(sy + 1, sx + 1) * (ay * ax)).astype(np.float32)


def revert_heatmap(heatmap, input_center, input_scale, img_shape):
    """utils.py:146-175 for a (K, h, w) map: -> (K, img_h, img_w)."""
    hm = np.asarray(heatmap, np.float32).transpose(1, 2, 0)
    m = get_warp_matrix(np.asarray(input_center).reshape(2), np.asarray(input_scale).reshape(2), 0, (hm.shape[1], hm.shape[0]), inv=True)
    return warp_affine_f32(hm, m, (img_shape[1], img_shape[0])).transpose(2, 0, 1)


def image_padding(centers, scales, ori_shape):
    """utils.py:70-87: [left, top, right, bottom] padding that keeps every activation window (+10 px) inside."""
    pad = np.zeros(4, np.int64)
    for c, s in zip(centers, scales):
        pad = np.maximum(pad, [int(max(s[0] / 2 - c[0] + 10, 0)), int(max(s[1] / 2 - c[1] + 10, 0)),
                               int(max(c[0] + s[0] / 2 - ori_shape[1] + 10, 0)), int(max(c[1] + s[1] / 2 - ori_shape[0] + 10, 0))])
    return pad


def merge_heatmaps(heatmaps, centers, scales, ori_shape):
    """utils.py:66-123: (merged maps on the image, merged maps on the padded image, padding)."""
    pad = image_padding(centers, scales, ori_shape)
    shape_p = (ori_shape[0] + pad[1] + pad[3], ori_shape[1] + pad[0] + pad[2])
    plain = [revert_heatmap(h, c, s, ori_shape) for h, c, s in zip(heatmaps, centers, scales)]
    padded = [revert_heatmap(h, np.asarray(c) + pad[:2], s, shape_p) for h, c, s in zip(heatmaps, centers, scales)]
    return np.max(plain, axis=0), np.max(padded, axis=0), pad


def posterior(heatmaps, presence_probs):
    """local_visualizer.py:827-837: maps normalised to sum 1, times the presence probability averaged over the instances."""
    heatmaps = np.asarray(heatmaps, np.float32)
    heatmaps = heatmaps / heatmaps.sum(axis=(1, 2), keepdims=True)
    return heatmaps * np.asarray(presence_probs, np.float32).reshape(-1, heatmaps.shape[0]).mean(axis=0)[:, None, None]
