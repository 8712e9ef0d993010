"""Key-point decoders, MI355X-native: drop-in for the reference's `metrics/pose_metrics.py:10-107`.

    decoder = GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)
    kps, max_val = decoder(heat_map, trans_inv)   # [B,J,2], [B,J,1]  (same shapes/dtypes/device as the reference)

Each call is ONE kernel launch (sp_decode_*), asynchronous on torch's current stream, no host sync, and the input
heat map is not modified.
"""
from __future__ import annotations

import torch

from .. import _lib


def _prep(heat_map, trans_inv=None):
    heat_map = _lib.require_cuda_f32(heat_map, "heat_map")
    if heat_map.dim() != 4:
        raise ValueError(f"heat_map must be [B,J,H,W], got {tuple(heat_map.shape)}")
    B = heat_map.shape[0]
    if trans_inv is not None:
        trans_inv = _lib.require_cuda_f32(trans_inv, "trans_inv")
        if tuple(trans_inv.shape) != (B, 2, 3):
            raise ValueError(f"trans_inv must be [{B},2,3], got {tuple(trans_inv.shape)}")
        _lib.same_device(heat_map, trans_inv)
    return heat_map, trans_inv


class BasicKeyPointDecoder(object):
    @staticmethod
    def heat_map_to_axis(heat_map: torch.Tensor):
        """pose_metrics.py:11-24 -> (coords [B,J,2] float, max_val [B,J,1])."""
        heat_map, _ = _prep(heat_map)
        B, J, H, W = heat_map.shape
        coords = torch.empty((B, J, 2), dtype=torch.float32, device=heat_map.device)
        max_val = torch.empty((B, J, 1), dtype=torch.float32, device=heat_map.device)
        if B == 0:                      # an empty batch (an image without detections): empty results, as the reference's torch ops give
            return coords, max_val
        _lib.check(_lib.lib().sp_heat_map_to_axis(_lib.ptr(heat_map), B, J, H, W, _lib.ptr(coords), _lib.ptr(max_val),
                                                  _lib.current_stream(heat_map.device)), "sp_heat_map_to_axis")
        return coords, max_val

    @torch.no_grad()
    def __call__(self, heat_map: torch.Tensor, trans_inv: torch.Tensor):
        """pose_metrics.py:26-52: argmax + 0.25 px sign shift + affine to image coordinates."""
        heat_map, trans_inv = _prep(heat_map, trans_inv)
        B, J, H, W = heat_map.shape
        kps = torch.empty((B, J, 2), dtype=torch.float32, device=heat_map.device)
        max_val = torch.empty((B, J, 1), dtype=torch.float32, device=heat_map.device)
        if B == 0:
            return kps, max_val
        _lib.check(_lib.lib().sp_decode_basic(_lib.ptr(heat_map), _lib.ptr(trans_inv), B, J, H, W, _lib.ptr(kps),
                                              _lib.ptr(max_val), _lib.current_stream(heat_map.device)), "sp_decode_basic")
        return kps, max_val


class GaussTaylorKeyPointDecoder(BasicKeyPointDecoder):
    def __init__(self, kernel_size: int = 11, num_joints: int = 17):
        if kernel_size % 2 != 1 or not (1 <= kernel_size <= 15):
            raise ValueError("kernel_size must be odd and <= 15")
        self.kernel_size = kernel_size
        self.num_joints = num_joints

    @torch.no_grad()
    def __call__(self, heat_map: torch.Tensor, trans_inv: torch.Tensor):
        """pose_metrics.py:62-107: blur -> rescale -> log -> 2nd-order Taylor refinement -> affine."""
        heat_map, trans_inv = _prep(heat_map, trans_inv)
        B, J, H, W = heat_map.shape
        if J != self.num_joints:
            # the reference's depthwise conv (groups=num_joints) would raise on a channel mismatch too
            raise ValueError(f"decoder built for {self.num_joints} joints, heat map has {J}")
        kps = torch.empty((B, J, 2), dtype=torch.float32, device=heat_map.device)
        max_val = torch.empty((B, J, 1), dtype=torch.float32, device=heat_map.device)
        if B == 0:
            return kps, max_val
        _lib.check(_lib.lib().sp_decode_gauss_taylor(_lib.ptr(heat_map), _lib.ptr(trans_inv), B, J, H, W, self.kernel_size,
                                                     _lib.ptr(kps), _lib.ptr(max_val), _lib.current_stream(heat_map.device)),
                   "sp_decode_gauss_taylor")
        return kps, max_val


class HeatMapAcc(object):
    """PCK-style training accuracy, drop-in for `metrics/pose_metrics.py:212-245`: arg-max of predictions and targets (HIP),
    then the per-joint hit rate in one small kernel.  Returns a 0-dim device tensor; nothing syncs with the host."""

    def __init__(self, distance_thresh=0.5, norm_frac=10.):
        self.distance_thresh = distance_thresh
        self.norm_frac = norm_frac

    @torch.no_grad()
    def __call__(self, predicts: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        """`mask` [B,J] (optional): equivalent to passing `predicts.mul(mask[..., None, None])`, `targets.mul(...)` as the solver
        does (ddp...:130-131) for masks of zeros and positive weights, without materialising the products."""
        preds, _ = BasicKeyPointDecoder.heat_map_to_axis(predicts)
        labels, _ = BasicKeyPointDecoder.heat_map_to_axis(targets)
        B, J, H, W = predicts.shape
        if mask is not None:
            mask = _lib.require_cuda_f32(mask, "mask")
            if tuple(mask.shape) != (B, J):
                raise ValueError(f"mask: expected [{B},{J}], got {tuple(mask.shape)}")
        acc = torch.empty((), dtype=torch.float32, device=predicts.device)
        _lib.check(_lib.lib().sp_heat_map_acc(_lib.ptr(preds), _lib.ptr(labels), _lib.ptr(mask), B, J, H, W, float(self.distance_thresh),
                                              float(self.norm_frac), _lib.ptr(acc), _lib.current_stream()), "sp_heat_map_acc")
        return acc


def kps_to_dict_(predicts: torch.Tensor, scores: torch.Tensor, img_ids, set_in_list: list):
    """`metrics/pose_metrics.py:172-179`: append one COCO result dict per person; `score = mean + max` of the per-joint maxima is
    computed for the whole batch by one launch (the reference does B `.item()` syncs)."""
    predicts = _lib.require_cuda_f32(predicts, "predicts")
    scores = _lib.require_cuda_f32(scores, "scores")
    B, J = predicts.shape[0], predicts.shape[1]
    if B == 0:
        return
    sc = torch.empty(B, dtype=torch.float32, device=predicts.device)
    _lib.check(_lib.lib().sp_pose_score(_lib.ptr(scores), B, J, _lib.ptr(sc), _lib.current_stream()), "sp_pose_score")
    flat = torch.cat([predicts, scores.reshape(B, J, 1)], dim=-1).reshape(B, -1).cpu().tolist()    # output formatting only
    for kp, s, img_id in zip(flat, sc.cpu().tolist(), img_ids):
        set_in_list.append({"image_id": img_id, "score": float(s), "category_id": 1, "keypoints": kp})
