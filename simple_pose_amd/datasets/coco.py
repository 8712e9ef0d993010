"""Input contract of the hot path: the normalisation half of the reference's `MSCOCO.collate_fn`
(`datasets/coco.py:124-148`), on the GPU.  COCO parsing / augmentation stay out of scope."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib

rgb_mean = [0.485, 0.456, 0.406]   # datasets/coco.py:10 (std is NOT applied: coco.py:134-136)


def normalize_crops(img_u8_bhwc_bgr: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,3] BGR crops on the GPU -> fp32 [B,3,H,W] RGB, `x / 255 - mean` (coco.py:136), the tensor `model(x)` takes.
    Ships 1 byte per value over PCIe instead of 4 and removes the per-sample numpy work from the dataloader."""
    t = img_u8_bhwc_bgr
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 4 and t.shape[-1] == 3):
        raise _lib.HipLibraryError("normalize_crops: expected a CUDA uint8 tensor [B,H,W,3]")
    t = t.contiguous()
    B, H, W, _ = t.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=t.device)
    if B == 0:
        return out
    mean = (ctypes.c_float * 3)(*rgb_mean)
    _lib.check(_lib.lib().sp_u8hwc_bgr_to_nchw_f32(_lib.ptr(t), _lib.ptr(out), B, H, W, mean, _lib.current_stream()),
               "sp_u8hwc_bgr_to_nchw_f32")
    return out
