"""Build-owned deterministic synthetic data: weights, inputs, joints.

Nothing here uses torch's or numpy's RNG streams: every tensor is a pure function of
``(seed, name, shape)`` through a splitmix64 counter hash, uniform values are exact
multiples of 2**-24 and the "normal" draws are Irwin-Hall(4) sums of those, so the
float64 intermediate arithmetic is exact and the generated fp32 tensors are
bit-identical on every IEEE-754 machine (this container, the GPU box).

The weight recipe is the *conditioned* init of SURVEY.md App. E: the reference's own
init (``nets/pose_resnet_dconv.py:180-189``, conv/deconv N(0, 0.001), BN 1/0) gives
heat maps of magnitude 1e-16 in eval mode, which the decoder's ``clamp(min=1e-10)``
(``metrics/pose_metrics.py:73``) flattens - useless for parity.  This recipe keeps
activations O(1) through all stages so the Taylor refinement actually fires.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
RGB_MEAN = (0.485, 0.456, 0.406)  # datasets/coco.py:10 (mean only, no std: coco.py:136)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = _splitmix64(np.array([(int(seed) << 32) ^ h], dtype=np.uint64))
    return k[0]


def uniform01(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values in [0,1), each an exact multiple of 2**-24."""
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(8) + np.uint64(lane)) & _MASK
        bits = _splitmix64(ctr ^ key)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def normal01(seed: int, name: str, n: int) -> np.ndarray:
    """Approximately N(0,1): Irwin-Hall sum of 4 uniforms, centred, unit variance (exact in fp64)."""
    s = uniform01(seed, name, n, 0)
    for lane in (1, 2, 3):
        s = s + uniform01(seed, name, n, lane)
    return (s - 2.0) * 1.7320508075688772  # var of sum of 4 U = 4/12 -> * sqrt(3)


def tensor_normal(seed, name, shape, std=1.0, mean=0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (normal01(seed, name, n) * std + mean).astype(np.float32).reshape(shape)


def tensor_uniform(seed, name, shape, lo=0.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (uniform01(seed, name, n) * (hi - lo) + lo).astype(np.float32).reshape(shape)


def conditioned_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...], str]], seed: int = 0,
                           transposed_prefixes: Tuple[str, ...] = ("deconv_layers",)) -> Dict[str, np.ndarray]:
    """Generate a conditioned state_dict for ``shapes`` = iterable of (key, shape, dtype-name).

    Rules:
      * conv weight [O,I,kh,kw]:     N(0, sqrt(2 / (I*kh*kw)))
      * transposed-conv weight [I,O,kh,kw] (keys under ``transposed_prefixes``): fan_in = I*kh*kw/4
      * conv bias:                   N(0, 0.1)
      * BN weight U(0.75,1.25) (x0.25 for every ``bn3``), BN bias N(0,0.1),
        running_mean N(0,0.1), running_var U(0.75,1.25), num_batches_tracked 0
    """
    out: Dict[str, np.ndarray] = {}
    shapes = list(shapes)
    four_d = {k for k, s, _ in shapes if len(s) == 4}
    for key, shape, dt in shapes:
        shape = tuple(int(v) for v in shape)
        leaf = key.rsplit(".", 1)[-1]
        parent = key.rsplit(".", 1)[0] if "." in key else ""
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif len(shape) == 4:
            is_t = any(key.startswith(p) for p in transposed_prefixes)
            fan_in = shape[0] * shape[2] * shape[3] / 4.0 if is_t else shape[1] * shape[2] * shape[3]
            out[key] = tensor_normal(seed, key, shape, std=float(np.sqrt(2.0 / fan_in)))
        elif leaf == "running_mean":
            out[key] = tensor_normal(seed, key, shape, std=0.1)
        elif leaf == "running_var":
            out[key] = tensor_uniform(seed, key, shape, 0.75, 1.25)
        elif leaf == "bias" and (parent + ".weight") in four_d:
            out[key] = tensor_normal(seed, key, shape, std=0.1)  # conv bias
        elif leaf == "bias":
            out[key] = tensor_normal(seed, key, shape, std=0.1)  # BN beta
        elif leaf == "weight":
            g = tensor_uniform(seed, key, shape, 0.75, 1.25)
            # damp the last BN of every residual block (Bottleneck.bn3; HRNet BasicBlock.bn2 inside `branches`) so that
            # activations stay O(1) through dozens of residual additions
            # ... and the BN of every HRNet fuse path (up to four branches are summed eight times over)
            if parent.endswith("bn3") or (".branches." in key and parent.endswith("bn2")) or ".fuse_layers." in key:
                g = (g * np.float32(0.25)).astype(np.float32)
            out[key] = g
        else:
            raise KeyError(f"no init rule for state_dict key {key!r} shape {shape}")
    return out


def state_dict_shapes(module) -> list:
    """(key, shape, dtype) triples of a torch module's state_dict."""
    return [(k, tuple(v.shape), str(v.dtype)) for k, v in module.state_dict().items()]


def load_conditioned(module, seed: int = 0):
    """Fill ``module`` (reference net or ours - identical keys) with the conditioned weights, strict."""
    import torch

    sd = conditioned_state_dict(state_dict_shapes(module), seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return module


def input_images(batch: int, seed: int = 0, h: int = 256, w: int = 192) -> np.ndarray:
    """fp32 NCHW [B,3,h,w] = U[0,1) - rgb_mean, mimicking datasets/coco.py:136 (range ~[-0.485, 0.594])."""
    x = np.empty((batch, 3, h, w), dtype=np.float32)
    for b in range(batch):
        x[b] = tensor_uniform(seed, f"input/{b}", (3, h, w))
    x -= np.asarray(RGB_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    return x


def trans_inv_batch(batch: int, seed: int | None = None) -> np.ndarray:
    """[B,2,3] fp32. seed None -> the x4 scaling [[4,0,0],[0,4,0]] of SURVEY.md 8(d); else random affine."""
    if seed is None:
        t = np.zeros((batch, 2, 3), dtype=np.float32)
        t[:, 0, 0] = 4.0
        t[:, 1, 1] = 4.0
        return t
    t = tensor_normal(seed, "trans_inv", (batch, 2, 3), std=1.5)
    t[:, 0, 0] += 4.0
    t[:, 1, 1] += 4.0
    t[:, :, 2] = tensor_uniform(seed, "trans_inv/t", (batch, 2), -50.0, 300.0)
    return t


def joints_batch(batch: int, joints: int = 17, seed: int = 0, w: int = 48, h: int = 64,
                 vis_p: float = 0.8) -> np.ndarray:
    """[B,J,3] fp32 (x,y,vis) in heat-map px: x~U(-4,w+4), y~U(-4,h+4), vis~Bernoulli(vis_p) (SURVEY 8d cfg 4)."""
    j = np.empty((batch, joints, 3), dtype=np.float32)
    j[..., 0] = tensor_uniform(seed, "joints/x", (batch, joints), -4.0, w + 4.0)
    j[..., 1] = tensor_uniform(seed, "joints/y", (batch, joints), -4.0, h + 4.0)
    j[..., 2] = (tensor_uniform(seed, "joints/v", (batch, joints)) < vis_p).astype(np.float32)
    return j
