"""Box -> crop geometry of the detector-driven path: the two helpers of the reference's `commons/joint_utils.py` that sit
next to the hot path - `box_to_center_scale` (:39-56) and `get_affine_transform` (:115-152).  A few float operations per person
on the host (they only produce the 2x3 matrices); the pixel work, `cv.warpAffine`, runs on the GPU
(`simple_pose_amd.datasets.naive_data.crop_boxes`).  Arithmetic types follow the reference step by step (float32 points,
float64 solve) so that the matrices agree to the last bits."""
from __future__ import annotations

import math

import numpy as np

_F32 = np.float32


def box_to_center_scale(x, y, w, h, aspect_ratio=1.0, scale_mult=1.25):
    """(center [2] float32, scale [2] float32): box centre; the box grown to `aspect_ratio` (= w/h of the network input) on its
    short side, then enlarged by `scale_mult` - unless the centre's x is the sentinel -1."""
    cx, cy = x + w * 0.5, y + h * 0.5
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    center = np.array([cx, cy], dtype=_F32)
    scale = np.array([w, h], dtype=_F32)
    return center, (scale * scale_mult if center[0] != -1 else scale)


def _solve_affine(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """cv.getAffineTransform(p, q): the 2x3 float64 map taking the three points p[i] to q[i].  Cramer's rule in Python floats
    (fixed operation order, so the matrix is bit-identical on every host; a LAPACK solve is not)."""
    (x0, y0), (x1, y1), (x2, y2) = [(float(a), float(b)) for a, b in p]
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    out = np.empty((2, 3), np.float64)
    for k in range(2):
        u0, u1, u2 = float(q[0][k]), float(q[1][k]), float(q[2][k])
        out[k, 0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        out[k, 1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        out[k, 2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
    return out


def _triangle(p0, p1) -> np.ndarray:
    """Three float32 points: p0, p1 (each rounded to float32 once), and the corner that makes a right angle at p1."""
    pts = np.zeros((3, 2), dtype=_F32)
    pts[0] = p0
    pts[1] = p1
    d = pts[0] - pts[1]
    pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=_F32)
    return pts


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=_F32)):
    """(trans, trans_inv), both 2x3 float64: `trans` maps image coordinates of the (center, scale) box, rotated by `rot`
    degrees, onto an `output_size` = (w, h) crop; `trans_inv` goes back (what the decoders take).  Only the box WIDTH sets the
    zoom (the reference's convention)."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    out_w, out_h = output_size[0], output_size[1]
    theta = math.pi * rot / 180
    sn, cs = np.sin(theta), np.cos(theta)
    half = scale[0] * -0.5                                       # the "up" vector of the box: (0, -w/2), rotated
    offset = scale * shift
    up = np.array([0 * cs - half * sn, 0 * sn + half * cs])       # float64
    src = _triangle(center + offset, center + up + offset)
    mid = np.array([out_w * 0.5, out_h * 0.5])
    dst = _triangle(mid, mid + np.array([0, out_w * -0.5], _F32))
    return _solve_affine(src, dst), _solve_affine(dst, src)
