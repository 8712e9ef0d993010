"""oracle/pose_oracle.py - TEST INFRASTRUCTURE.  ctypes front-end of oracle/pose_oracle.c (numpy in/out)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpose_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pose_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", _HERE, "-f", os.path.join(_HERE, "Makefile")] + (["-B"] if force else []),
                       check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.sp_oracle_masked_mse.restype = ctypes.c_double
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def blur_kernel(ks: int = 11) -> np.ndarray:
    k = np.empty((ks, ks), np.float32)
    lib().sp_oracle_blur_kernel(ctypes.c_int(ks), _p(k))
    return k


def heat_map_to_axis(heat):
    heat = _f32(heat)
    B, J, H, W = heat.shape
    coords = np.empty((B, J, 2), np.float32)
    mv = np.empty((B, J, 1), np.float32)
    lib().sp_oracle_heat_map_to_axis(_p(heat), B, J, H, W, _p(coords), _p(mv))
    return coords, mv


def decode_gauss_taylor(heat, trans_inv, kernel_size: int = 11):
    heat, trans_inv = _f32(heat), _f32(trans_inv)
    B, J, H, W = heat.shape
    assert trans_inv.shape == (B, 2, 3)
    kps = np.empty((B, J, 2), np.float32)
    mv = np.empty((B, J, 1), np.float32)
    lib().sp_oracle_decode_gauss_taylor(_p(heat), _p(trans_inv), B, J, H, W, kernel_size, _p(kps), _p(mv))
    return kps, mv


def decode_basic(heat, trans_inv):
    heat, trans_inv = _f32(heat), _f32(trans_inv)
    B, J, H, W = heat.shape
    kps = np.empty((B, J, 2), np.float32)
    mv = np.empty((B, J, 1), np.float32)
    lib().sp_oracle_decode_basic(_p(heat), _p(trans_inv), B, J, H, W, _p(kps), _p(mv))
    return kps, mv


def encode_refine(joints, sigma: float = 2.0, shape=(48, 64)):
    """joints [B,J,3] or [J,3]; shape=(W,H) as in the reference; returns targets [.., J,H,W], weights [.., J]."""
    joints = _f32(joints)
    single = joints.ndim == 2
    if single:
        joints = joints[None]
    B, J, _ = joints.shape
    W, H = shape
    t = np.empty((B, J, H, W), np.float32)
    w = np.empty((B, J), np.float32)
    lib().sp_oracle_encode_refine(_p(joints), B, J, H, W, ctypes.c_float(sigma), _p(t), _p(w))
    return (t[0], w[0]) if single else (t, w)


def encode_basic(joints, sigma: float = 2.0, shape=(48, 64), stride: int = 4):
    joints = _f32(joints)
    single = joints.ndim == 2
    if single:
        joints = joints[None]
    B, J, _ = joints.shape
    W, H = shape
    t = np.empty((B, J, H, W), np.float32)
    w = np.empty((B, J), np.float32)
    lib().sp_oracle_encode_basic(_p(joints), B, J, H, W, ctypes.c_float(sigma), int(stride), _p(t), _p(w))
    return (t[0], w[0]) if single else (t, w)


def masked_mse(pred, target, mask, want_grad: bool = False):
    pred, target, mask = _f32(pred), _f32(target), _f32(mask)
    B, J, H, W = pred.shape
    g = np.empty_like(pred) if want_grad else None
    loss = lib().sp_oracle_masked_mse(_p(pred), _p(target), _p(mask), B, J, H * W, _p(g) if want_grad else None)
    return (np.float32(loss), g) if want_grad else np.float32(loss)


def heat_map_acc(pred, target, distance_thresh: float = 0.5, norm_frac: float = 10.0) -> np.float32:
    """HeatMapAcc.__call__, metrics/pose_metrics.py:212-245 (numpy restatement on top of heat_map_to_axis)."""
    p, _ = heat_map_to_axis(pred)
    l, _ = heat_map_to_axis(target)
    H, W = pred.shape[-2:]
    norm = np.array([W, H], np.float32) / np.float32(norm_frac)            # :225-228
    valid = (l[..., 0] > 1) & (l[..., 1] > 1)                              # :229
    d = np.sqrt((((p / norm) - (l / norm)) ** 2).sum(-1, dtype=np.float32))  # :230
    acc_sum, cnt = np.float32(0), 0
    for j in range(d.shape[1]):                                           # :234-241
        v = valid[:, j]
        if v.sum() < 1:
            continue
        acc_sum += np.float32((d[v, j] < distance_thresh).sum()) / np.float32(v.sum())
        cnt += 1
    return np.float32(acc_sum / cnt) if cnt > 0 else np.float32(0)


def normalize_crops(img_u8_bhwc_bgr, mean=(0.485, 0.456, 0.406)):
    """datasets/coco.py:136 + :137: (img[..., ::-1].astype(float32) / 255.0 - rgb_mean), HWC -> CHW."""
    x = img_u8_bhwc_bgr[..., ::-1].astype(np.float32) / np.float32(255.0) - np.asarray(mean, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pose_rescore(kps, box_score, in_vis_thre: float = 0.2) -> np.ndarray:
    """eval.py:166-174: box_score * mean(visible key-point scores); kps [P,J,3] (x, y, score)."""
    kps, box_score = _f64(kps), _f64(box_score)
    P, J, _ = kps.shape
    out = np.empty(P, np.float64)
    lib().sp_oracle_pose_rescore(_p(kps), _p(box_score), P, J, ctypes.c_double(in_vis_thre), _p(out))
    return out


def oks_nms(kps, scores, areas, thresh, sigmas=None, in_vis_thresh=None):
    """datasets/naive_data.py:153-173 on one image's persons -> list of kept indices in pick order."""
    kps, scores, areas = _f64(kps), _f64(scores), _f64(areas)
    N, J, _ = kps.shape
    keep = np.empty(max(N, 1), np.int32)
    sg = _f64(sigmas) if sigmas is not None else None
    n = lib().sp_oracle_oks_nms(_p(kps), _p(scores), _p(areas), N, J, _p(sg) if sg is not None else None, ctypes.c_double(thresh),
                                ctypes.c_double(-1.0 if in_vis_thresh is None else in_vis_thresh), _p(keep))
    return keep[:n].tolist()


def pose_score(max_val) -> np.ndarray:
    """kps_to_dict_'s per-person score, metrics/pose_metrics.py:176: sc.mean() + sc.max(); max_val [B,J] or [B,J,1]."""
    m = _f32(max_val).reshape(len(max_val), -1)
    out = np.empty(m.shape[0], np.float32)
    lib().sp_oracle_pose_score(_p(m), m.shape[0], m.shape[1], _p(out))
    return out


def warp_affine_u8c3(img, M, dsize):
    """cv.warpAffine(img, M, dsize=(w, h), flags=cv.INTER_LINEAR) for uint8 HxWx3 (restated OpenCV arithmetic; see pose_oracle.c)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, C = img.shape
    assert C == 3
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(6)
    ow, oh = int(dsize[0]), int(dsize[1])
    out = np.empty((oh, ow, 3), np.uint8)
    lib().sp_oracle_warp_affine_u8c3(_p(img), H, W, _p(M), _p(out), oh, ow)
    return out


def get_affine_transform_3pt(src, dst):
    """cv.getAffineTransform(src[3,2] float32, dst[3,2] float32) -> 2x3 float64.  OpenCV solves the 6x6 system by LU in float64;
    here Cramer's rule in plain Python floats (IEEE double operations in a fixed order): the same map to ~1e-16 relative and,
    unlike a LAPACK call, bit-reproducible on every machine (the GPU box's host CPU is not the build container's)."""
    (x0, y0), (x1, y1), (x2, y2) = [(float(a), float(b)) for a, b in np.asarray(src, np.float64)]
    d = np.asarray(dst, np.float64)
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    rows = []
    for k in range(2):
        u0, u1, u2 = float(d[0, k]), float(d[1, k]), float(d[2, k])
        a = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        b = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        c = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
        rows.append([a, b, c])
    return np.array(rows, np.float64)
