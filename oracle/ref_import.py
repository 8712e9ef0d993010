"""Import the upstream reference (liangheming/simple_pose at /root/reference) - BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE.  Used by oracle/gen_golden.py (fixture generation) and by the optional
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, i.e. on the GPU box).
The reference never travels: only the numeric outputs it produces are committed (tests/golden/).

Two in-memory shims are needed:
  * ``cv2`` and ``pycocotools`` are not installed -> stub modules; ``cv2.getGaussianKernel(k, 0)`` is
    replaced by its closed form (sigma = 0.3*((k-1)*0.5-1)+0.8, normalised float64 column vector).
  * ``metrics/pose_metrics.py:102`` (``valid_mask[valid_mask] = ...``) raises on torch >= 2.x because the
    index aliases the destination; the module source is exec'd with the index ``.clone()``d - identical
    semantics.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("SIMPLE_POSE_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "metrics", "pose_metrics.py"))


def _gaussian_kernel(ksize, sigma, ktype=None):
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).reshape(ksize, 1)


def _install_stubs():
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.getGaussianKernel = _gaussian_kernel
        cv2.setNumThreads = lambda n: None
        cv2.INTER_LINEAR = 1
        # restated OpenCV primitives (oracle/pose_oracle.*): let the reference's crop path run without opencv-python
        from . import pose_oracle as _po
        cv2.getAffineTransform = lambda src, dst: _po.get_affine_transform_3pt(src, dst)
        cv2.warpAffine = lambda img, M, dsize, flags=1: _po.warp_affine_u8c3(img, M, dsize)
        cv2.COLOR_GRAY2BGR = 8
        sys.modules["cv2"] = cv2
    if "pycocotools" not in sys.modules:
        pkg = types.ModuleType("pycocotools")
        coco = types.ModuleType("pycocotools.coco")
        coco.COCO = object
        ce = types.ModuleType("pycocotools.cocoeval")
        ce.COCOeval = object
        pkg.coco, pkg.cocoeval = coco, ce
        sys.modules.update({"pycocotools": pkg, "pycocotools.coco": coco, "pycocotools.cocoeval": ce})


_cache = {}


def load():
    """Returns a namespace with the reference modules: dconv, duc, hrnet, pose_metrics, transforms."""
    if "ns" in _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # /root/reference must stay untouched
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    ns.dconv = importlib.import_module("nets.pose_resnet_dconv")
    ns.duc = importlib.import_module("nets.pose_resnet_duc")
    ns.hrnet = importlib.import_module("nets.pose_hrnet")
    ns.transforms = importlib.import_module("commons.transforms")
    path = os.path.join(REFERENCE_ROOT, "metrics", "pose_metrics.py")
    with open(path, "r", encoding="utf-8") as fh:
        src = fh.read()
    old = "valid_mask[valid_mask] = derivative_valid_mask"
    assert src.count(old) == 1, "reference decoder changed; re-check the torch-2.x shim"
    src = src.replace(old, "valid_mask[valid_mask.clone()] = derivative_valid_mask")
    mod = types.ModuleType("metrics.pose_metrics")
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    ns.pose_metrics = mod
    ns.hrnet_w32_yaml = os.path.join(REFERENCE_ROOT, "nets", "hrnet_w32.yaml")
    _cache["ns"] = ns
    return ns
