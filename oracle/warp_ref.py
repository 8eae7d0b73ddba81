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
