"""Heat-map target encoders, MI355X-native: drop-in for `get_heat_map` of the reference's
`commons/transforms.py` (RefineSimpleTransform :167-191 - the one the dataset uses - and BasicSimpleTransform :80-116).

The reference encodes one sample at a time in numpy inside a dataloader worker (3.7 ms / image).  Here the same
function also takes a whole batch `[B,J,3]` that already lives on the GPU and returns device tensors (one launch);
a numpy `[J,3]` argument keeps the reference's numpy-in / numpy-out contract (upload, launch, download).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def _run(fn_name, joints, sigma, shape, stride=None):
    is_np = isinstance(joints, np.ndarray)
    j = torch.from_numpy(np.ascontiguousarray(joints, dtype=np.float32)).cuda() if is_np else joints
    j = _lib.require_cuda_f32(j, "joints")
    single = j.dim() == 2
    if single:
        j = j[None]
    if j.dim() != 3 or j.shape[-1] != 3:
        raise ValueError(f"joints must be [J,3] or [B,J,3], got {tuple(joints.shape)}")
    j = j.contiguous()
    B, J, _ = j.shape
    W, H = int(shape[0]), int(shape[1])  # the reference passes shape=(w, h) and returns [J, h, w]
    targets = torch.empty((B, J, H, W), dtype=torch.float32, device=j.device)
    weights = torch.empty((B, J), dtype=torch.float32, device=j.device)
    lib = _lib.lib()
    if B == 0:
        rc = 0                          # empty batch: empty targets / weights, nothing to launch
    elif stride is None:
        rc = lib.sp_encode_gauss_refine(_lib.ptr(j), B, J, H, W, float(sigma), _lib.ptr(targets), _lib.ptr(weights),
                                        _lib.current_stream())
    else:
        rc = lib.sp_encode_gauss_basic(_lib.ptr(j), B, J, H, W, float(sigma), int(stride), _lib.ptr(targets),
                                       _lib.ptr(weights), _lib.current_stream())
    _lib.check(rc, fn_name)
    if single:
        targets, weights = targets[0], weights[0]
    if is_np:
        return targets.cpu().numpy(), weights.cpu().numpy()
    return targets, weights


class BasicSimpleTransform(object):
    @staticmethod
    def get_heat_map(joints, sigma=2.0, shape=(48, 64), stride=4):
        """transforms.py:80-116: joints in INPUT px; centre quantised to int(j/stride+0.5); 13x13 truncated patch."""
        return _run("sp_encode_gauss_basic", joints, sigma, shape, stride)


class RefineSimpleTransform(object):
    @staticmethod
    def get_heat_map(joints, sigma=2.0, shape=(48, 64)):
        """transforms.py:167-191: joints in heat-map px (un-quantised); full-map Gaussian; weight 0 if the 3-sigma box
        misses the map."""
        return _run("sp_encode_gauss_refine", joints, sigma, shape)
