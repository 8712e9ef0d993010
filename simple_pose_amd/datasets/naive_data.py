"""Post-decode filtering of the detector-driven inference path: per-image pose rescoring and greedy OKS-NMS on the GPU.

Mirrors `datasets/naive_data.py:153-173` (`oks_nms`; `oks_iou` :120-150) and the loop of `eval.py:153-197`
(`temp_read_in_and_filter`).  The reference does this in numpy float64 on the host after writing the predictions to a JSON
file; here the decoder's output stays on the device and every image of the batch is filtered by one launch."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib

P = _lib.ptr


def _cuda_f64(a, name: str) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise _lib.HipLibraryError(f"{name}: tensor is on {a.device}; simple_pose_amd runs on the MI355X only (no CPU fallback)")
        return a.to(torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def _segments(img_ids: Sequence) -> tuple:
    """Group rows by image id in first-appearance order (the `defaultdict(list)` of eval.py:160-162).  Host-side bookkeeping."""
    first, rows = {}, []
    for i, iid in enumerate(img_ids):
        rows.append((first.setdefault(iid, len(first)), i))
    perm = [i for _, i in sorted(rows)]                      # stable: keeps the within-image order
    counts = np.bincount([g for g, _ in rows], minlength=len(first))
    seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return list(first.keys()), perm, seg


def oks_nms_batch(kps: torch.Tensor, scores: torch.Tensor, areas: torch.Tensor, seg: torch.Tensor, max_group: int, thresh: float,
                  sigmas=None, in_vis_thresh: Optional[float] = None):
    """kps [P,J,3] f64, scores [P] f64, areas [P] f64 (CUDA), seg int32 [G+1] (CUDA) -> (keep int32 [P], keep_count int32 [G])."""
    Pn, J, _ = kps.shape
    G = seg.numel() - 1
    keep = torch.empty(Pn, dtype=torch.int32, device=kps.device)
    cnt = torch.empty(G, dtype=torch.int32, device=kps.device)
    sg = None
    if sigmas is not None:
        sg = np.ascontiguousarray(sigmas, dtype=np.float64)
        if sg.shape != (J,):
            raise ValueError(f"sigmas: expected {J} values")
    _lib.check(_lib.lib().sp_oks_nms(P(kps), P(scores), P(areas), P(seg), G, int(max_group), J, sg.ctypes.data if sg is not None else None,
                                     float(thresh), -1.0 if in_vis_thresh is None else float(in_vis_thresh), P(keep), P(cnt),
                                     _lib.current_stream()), "sp_oks_nms")
    return keep, cnt


def oks_nms(kps, scores, areas, thresh, sigmas=None, in_vis_thresh=None) -> List[int]:
    """Drop-in for the reference's `oks_nms` (one image): kps [N,J,3], scores [N], areas [N] (numpy or CUDA tensors) -> indices
    kept, in pick order."""
    k, s, a = _cuda_f64(kps, "kps"), _cuda_f64(scores, "scores"), _cuda_f64(areas, "areas")
    n = k.shape[0]
    if n == 0:
        return []
    seg = torch.tensor([0, n], dtype=torch.int32, device=k.device)
    keep, cnt = oks_nms_batch(k, s, a, seg, n, thresh, sigmas, in_vis_thresh)
    return keep[: int(cnt.item())].tolist()


def filter_poses(kps: torch.Tensor, box_score, area, img_ids: Sequence, in_vis_thre: float = 0.2, oks_thre: float = 0.9) -> List[dict]:
    """eval.py:153-197 for one batch of decoded persons: kps [P,J,3] fp32 CUDA = cat(predicts, max_val) as eval.py:138 builds it,
    box_score / area [P] (detector confidence, crop area), img_ids [P] -> the COCO result dicts that survive OKS-NMS, image by
    image in first-appearance order, each image's poses in pick order."""
    kps = _lib.require_cuda_f32(kps, "kps")
    Pn, J, _ = kps.shape
    if Pn == 0:
        return []
    ids, perm, seg_h = _segments(list(img_ids))
    dev = kps.device
    perm_t = torch.tensor(perm, dtype=torch.int64, device=dev)
    kps_g = kps.index_select(0, perm_t).contiguous()                       # rows of one image made contiguous (layout glue)
    box = _cuda_f64(box_score, "box_score").index_select(0, perm_t).contiguous()
    # float(info.area): the area is a float32 product (naive_data.py:57) widened to double
    ar = (area if isinstance(area, torch.Tensor) else torch.from_numpy(np.asarray(area, np.float32))).to(dev).to(torch.float32)
    ar = ar.to(torch.float64).index_select(0, perm_t).contiguous()
    kps64 = torch.empty((Pn, J, 3), dtype=torch.float64, device=dev)
    score = torch.empty(Pn, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().sp_pose_rescore(P(kps_g), P(box), Pn, J, float(in_vis_thre), P(kps64), P(score), _lib.current_stream()),
               "sp_pose_rescore")
    seg = torch.from_numpy(seg_h).to(dev)
    keep, cnt = oks_nms_batch(kps64, score, ar, seg, int(np.diff(seg_h).max()), oks_thre)
    keep_h, cnt_h, score_h, kps_h = keep.cpu().numpy(), cnt.cpu().numpy(), score.cpu().numpy(), kps64.cpu().numpy()
    out = []
    for g, iid in enumerate(ids):
        if cnt_h[g] < 0:
            raise _lib.HipLibraryError(f"image {iid}: more than 2048 persons")
        for r in keep_h[seg_h[g]: seg_h[g] + cnt_h[g]]:
            out.append({"image_id": iid, "score": float(score_h[r]), "category_id": 1, "keypoints": kps_h[r].reshape(-1).tolist()})
    return out


def crop_boxes(img: torch.Tensor, boxes, input_shape=(192, 256), output_shape=(48, 64)):
    """`BasicTransform.__call__` (datasets/naive_data.py:33-56) for every detected box of one image, on the GPU: uint8 BGR image
    [H,W,3] (CUDA) + boxes [N,4] (x1,y1,x2,y2; host) -> (crops uint8 [N,h,w,3] ready for `datasets.coco.normalize_crops`,
    trans_inv float32 [N,2,3] for the decoder, centers [N,2], scales [N,2], areas [N]).  One launch for all N warps."""
    from ..commons.joint_utils import box_to_center_scale, get_affine_transform
    if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[-1] == 3):
        raise _lib.HipLibraryError("crop_boxes: expected a CUDA uint8 image [H,W,3]")
    img = img.contiguous()
    boxes = np.asarray(boxes).reshape(-1, 4)        # dtype kept: the box arithmetic runs in the caller's precision, as the reference's does
    n = boxes.shape[0]
    w_h_ratio = input_shape[0] / input_shape[1]
    m_fwd = np.empty((n, 2, 3), np.float64)
    tinv = np.empty((n, 2, 3), np.float32)
    centers, scales = np.empty((n, 2), np.float32), np.empty((n, 2), np.float32)
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        center, scale = box_to_center_scale(x1, y1, x2 - x1, y2 - y1, w_h_ratio)
        m_fwd[i], _ = get_affine_transform(center, scale, 0, input_shape)
        _, tinv[i] = get_affine_transform(center, scale, 0, output_shape)
        centers[i], scales[i] = center, scale
    crops = torch.empty((n, input_shape[1], input_shape[0], 3), dtype=torch.uint8, device=img.device)
    if n == 0:                          # no detections in this image: empty crops / matrices, nothing to launch
        return crops, torch.from_numpy(tinv).to(img.device), centers, scales, scales[:, 0] * scales[:, 1]
    _lib.check(_lib.lib().sp_warp_affine_u8c3(P(img), img.shape[0], img.shape[1], m_fwd.ctypes.data, n, P(crops), input_shape[1], input_shape[0],
                                              _lib.current_stream()), "sp_warp_affine_u8c3")
    return crops, torch.from_numpy(tinv).to(img.device), centers, scales, scales[:, 0] * scales[:, 1]
