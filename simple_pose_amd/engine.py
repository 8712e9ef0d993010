"""Host-side lowering of the reference networks onto libsimple_pose_hip.so.

A network is compiled once into a `Program`: a flat list of launches (sp_conv2d_fwd, sp_maxpool3x3s2_nhwc, ...)
over named NHWC fp32 activation buffers, with weights pre-packed into the kernel's [n_pad][k_pad] layout and
eval-mode BatchNorm folded into a per-channel (scale, shift) epilogue.  Python/PyTorch is glue only: it owns the
device memory and the stream; every FLOP runs in the HIP library.

Reference behaviour being lowered (file:line relative to liangheming/simple_pose):
  nets/pose_resnet_dconv.py:251-265  ResNet._forward_impl        -> resnet_program(head="dconv")
  nets/pose_resnet_dconv.py:112-133  Bottleneck.forward          -> _bottleneck()
  nets/pose_resnet_dconv.py:230-249  _make_deconv_layer          -> deconv_k4s2p1 launches (4 output phases)
  nets/pose_resnet_duc.py:227-232, nets/commons.py:21-43  DUC    -> conv3x3 with fused PixelShuffle epilogue
"""
from __future__ import annotations

import ctypes
import os

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, SP_CONV_BF16, SP_CONV_OUT_NCHW, SP_CONV_PIXEL_SHUFFLE, SP_CONV_RELU

BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------
# weight packing: device kernels behind the C ABI (csrc/pack.hip)
# ------------------------------------------------------------------------------------------------
def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def n_pad_for(c_out: int) -> int:
    """Packed row count: the kernel's N tile must divide it (128 / 64 / 32 wide tiles)."""
    if c_out >= 128:
        return _round_up(c_out, 128)
    if c_out > 32:
        return _round_up(c_out, 64)
    return 32


class HipPacker:
    """Reference-layout parameters -> kernel layouts ON THE GPU through the C ABI (sp_pack_conv_weights, sp_pack_deconv_k4s2p1,
    sp_fold_bn, sp_conv_packed_dims): the same entry points a maintainer binding include/simple_pose_hip.h would call.  Tensors must
    live on the GPU; there is no host path (tests/desc_interp.TorchPacker restates the layouts in torch for the CPU-only host-logic
    tests and is never imported by the package)."""

    @staticmethod
    def _dims(c_out: int, k: int, bf16: bool) -> Tuple[int, int]:
        n_pad, k_pad = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().sp_conv_packed_dims(c_out, k, int(bf16), ctypes.byref(n_pad), ctypes.byref(k_pad)), "sp_conv_packed_dims")
        return n_pad.value, k_pad.value

    def conv(self, w: torch.Tensor, *, c_in_pad: Optional[int] = None, taps_w_pad: Optional[int] = None, pixel_shuffle: bool = False,
             pair_s0: int = -1, bf16: bool = False) -> Tuple[torch.Tensor, int, int, int, int]:
        """Conv2d weight [O,I,kh,kw] -> packed [n_pad, k_pad] with K ordered (ky, kx, c), c fastest.
        Returns (packed, taps_h, taps_w, c_in_packed, k_pad)."""
        w = _lib.require_cuda_f32(w.detach(), "conv weight")
        O, I, kh, kw = w.shape
        ci, tw = c_in_pad or I, taps_w_pad or kw
        n_pad, k_pad = self._dims(O, kh * tw * ci, bf16)
        out = torch.empty((n_pad, k_pad), dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
        _lib.check(_lib.lib().sp_pack_conv_weights(_lib.ptr(w), O, I, kh, kw, ci, tw, int(pixel_shuffle), pair_s0, n_pad, k_pad, _lib.ptr(out),
                                                   int(bf16), _lib.current_stream()), "sp_pack_conv_weights")
        return out, kh, tw, ci, k_pad

    def grouped(self, w: torch.Tensor, groups: int, panel: int, *, bf16: bool = False) -> torch.Tensor:
        """nn.Conv2d(groups = g) weight [O, O / g, kh, kw] -> block-diagonal panels [O, kh * kw * panel] (sp_pack_conv_weights_grouped)."""
        w = _lib.require_cuda_f32(w.detach(), "grouped conv weight")
        O, cpg, kh, kw = w.shape
        out = torch.empty((O, kh * kw * panel), dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
        _lib.check(_lib.lib().sp_pack_conv_weights_grouped(_lib.ptr(w), O, groups, kh, kw, panel, _lib.ptr(out), int(bf16), _lib.current_stream()),
                   "sp_pack_conv_weights_grouped")
        return out

    def deconv(self, w: torch.Tensor, *, bf16: bool = False) -> Tuple[torch.Tensor, int]:
        """ConvTranspose2d(k=4, s=2, p=1) weight [I,O,4,4] -> ([4 * n_pad, 4*I], n_pad): one 2x2-tap slab per output phase."""
        w = _lib.require_cuda_f32(w.detach(), "deconv weight")
        I, O, kh, kw = w.shape
        assert (kh, kw) == (4, 4) and I % 32 == 0
        n_pad = n_pad_for(O)
        out = torch.empty((4 * n_pad, 4 * I), dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
        _lib.check(_lib.lib().sp_pack_deconv_k4s2p1(_lib.ptr(w), I, O, n_pad, _lib.ptr(out), int(bf16), _lib.current_stream()),
                   "sp_pack_deconv_k4s2p1")
        return out, n_pad

    def fold_bn(self, weight, bias, running_mean, running_var, eps: float = BN_EPS, pixel_shuffle: bool = False):
        """Eval-mode BatchNorm2d as y = x*scale + shift (ATen's factoring: alpha = w/sqrt(var+eps), beta = b - mean*alpha)."""
        f = lambda t, n: _lib.require_cuda_f32(t.detach(), n)
        mean, var = f(running_mean, "running_mean"), f(running_var, "running_var")
        C = mean.numel()
        scale, shift = torch.empty(C, dtype=torch.float32, device=mean.device), torch.empty(C, dtype=torch.float32, device=mean.device)
        _lib.check(_lib.lib().sp_fold_bn(_lib.ptr(f(weight, "bn weight")), _lib.ptr(f(bias, "bn bias")), _lib.ptr(mean), _lib.ptr(var), C, eps,
                                         int(pixel_shuffle), _lib.ptr(scale), _lib.ptr(shift), _lib.current_stream()), "sp_fold_bn")
        return scale, shift

    def bias(self, b: torch.Tensor) -> torch.Tensor:
        return _lib.require_cuda_f32(b.detach(), "bias")


# ------------------------------------------------------------------------------------------------
# program
# ------------------------------------------------------------------------------------------------
@dataclass
class Op:
    kind: str                    # "conv" | "bb32" / "bb64" (fused BasicBlock, 32 / 64 channels) | "bneck64" (fused Bottleneck) | "stem7" / "hstem" (fused ResNet / HRNet stem) | "htrans" (HRNet transition1) | "maxpool" | ...
    src: str
    dst: str
    res: Optional[str] = None
    desc: Optional[ConvDesc] = None
    w: Optional[torch.Tensor] = None
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    args: tuple = ()
    flops: int = 0               # algorithmic FLOPs per image (2*MACs, real taps/channels only)
    name: str = ""
    lane: int = 0                # HIP stream the op is issued on (0 = the caller's stream); independent branches get their own
    direct: bool = False         # conv: use sp_conv3x3_direct (bf16 3x3, 32 -> 32 channels) instead of the implicit GEMM; same bits
    dst2: Optional[str] = None   # second output ("htrans": HRNet's transition1 writes the high- and the half-resolution branch in one launch)

    def writes(self) -> Tuple[str, ...]:
        return (self.dst,) + ((self.dst2,) if self.dst2 else ())

    def extra_inputs(self) -> Tuple[str, ...]:
        """Input buffers beyond src / res (the SELayer's gate logits; the further terms of a multi-term fuse)."""
        if self.kind == "se_gate":
            return (self.args[2],)
        if self.kind == "upsample_add_n":
            return tuple(self.args[4])
        return ()

    def reads(self) -> Tuple[str, ...]:
        return tuple(n for n in (self.src, self.res) + self.extra_inputs() if n)


@dataclass
class Program:
    ops: List[Op] = field(default_factory=list)
    shapes: Dict[str, Tuple[int, int, int]] = field(default_factory=dict)  # buffer -> (H, W, C) NHWC per image
    out_name: str = "heat"
    out_shape: Tuple[int, int, int] = (17, 64, 48)                          # NCHW per image
    _pools: Dict[int, Dict[str, torch.Tensor]] = field(default_factory=dict)
    tuned_for_batch: int = 0
    dtype: str = "fp32"                                                     # activation / weight dtype: "fp32" | "bf16"
    multi_stream: bool = True                                               # honour Op.lane (False: everything on the caller's stream)
    _sync: Dict[Tuple[int, str], tuple] = field(default_factory=dict)       # per (batch, device): cross-lane wait / record plan
    _streams: Dict[str, list] = field(default_factory=dict)
    _events: Dict[str, dict] = field(default_factory=dict)

    # -- buffer planning: greedy reuse of dead activations (keeps the working set small for L2 / MALL) --
    MAX_POOLS = 2     # activation pools kept alive (distinct batch sizes / devices); older ones are dropped with their sync plans

    def _alloc(self, batch: int, device, slot: int = 0) -> Dict[str, torch.Tensor]:
        """Activation pool of (batch, device).  `slot` > 0: a further, independent pool of the same shape (engine.InterleavedForward runs
        consecutive batches on different streams: each needs its own activations); those are owned by whoever asked and never evicted."""
        key = (batch, str(device)) if slot == 0 else (batch, str(device), slot)
        if key in self._pools:
            if slot == 0:
                self._pools[key] = self._pools.pop(key)      # most recently used last
            return self._pools[key]
        if slot != 0:
            # a slot holds ONE batch shape at a time: a caller whose batch size varies (detector-driven person counts) would otherwise
            # pile up `depth` full activation pools per distinct size until close()
            for old in [k for k in self._pools if len(k) == 3 and k[1] == str(device) and k[2] == slot and k[0] != batch]:
                del self._pools[old]
        while slot == 0 and sum(len(k) == 2 for k in self._pools) >= self.MAX_POOLS:   # a detector-driven caller sees a new person count per
            old = next(k for k in self._pools if len(k) == 2)                          # image: do not let every batch size keep a full pool forever
            del self._pools[old]
            self._sync.pop(old, None)
        last_use: Dict[str, int] = {}
        for i, op in enumerate(self.ops):
            for nm in (op.src, op.res) + op.writes() + op.extra_inputs():
                if nm:
                    last_use[nm] = i
        free: Dict[int, List[torch.Tensor]] = {}
        bufs: Dict[str, torch.Tensor] = {}
        for i, op in enumerate(self.ops):
            for dn in op.writes():
                if dn not in bufs and dn != self.out_name:
                    h, w, c = self.shapes[dn]
                    n = batch * h * w * c
                    pool = free.get(n)
                    bufs[dn] = pool.pop() if pool else torch.empty(
                        n, dtype=torch.bfloat16 if self.dtype == "bf16" else torch.float32, device=device)
            for nm in (op.src, op.res) + op.extra_inputs():
                if nm and nm in bufs and last_use[nm] == i and nm != "input":
                    free.setdefault(bufs[nm].numel(), []).append(bufs[nm])
        self._pools[key] = bufs
        return bufs

    # -- cross-lane ordering: which earlier ops (on OTHER lanes) an op has to wait for -----------------------------------
    def _plan_sync(self, batch: int, device, slot: int = 0) -> tuple:
        """Ops of one lane are ordered by their stream.  Across lanes an op waits (HIP event) for: the producers of what it reads
        (RAW), and every earlier reader / writer of the STORAGE it writes (WAR / WAW - the buffer planner hands dead storage to
        later ops, and on another lane "later" is no longer implied).  Per (waiting lane, signalling lane) only the latest op is
        kept.  Returns (waits per op, ops that record an event, last op per lane)."""
        key = (batch, str(device))
        if key in self._sync:
            return self._sync[key]
        bufs = self._alloc(batch, device, slot)             # (the slot's own pool: the aliasing structure is the same for every slot)
        store = lambda name: bufs[name].data_ptr() if name in bufs else ("@" + name)     # "input" / output: their own storage
        last_writer: Dict[str, int] = {}
        touched: Dict[object, List[int]] = {}
        waits: List[List[int]] = []
        for i, op in enumerate(self.ops):
            need: Dict[int, int] = {}                       # signalling lane -> latest op index
            cand = [last_writer[n] for n in op.reads() if n in last_writer] + [j for dn in op.writes() for j in touched.get(store(dn), [])]
            for j in cand:
                lj = self.ops[j].lane
                if lj != op.lane and need.get(lj, -1) < j:
                    need[lj] = j
            waits.append(sorted(need.values()))
            for dn in op.writes():
                last_writer[dn] = i
            for n in op.reads() + op.writes():
                touched.setdefault(store(n), []).append(i)
        records = sorted({j for w in waits for j in w})
        tails: Dict[int, int] = {}
        for i, op in enumerate(self.ops):
            tails[op.lane] = i
        self._sync[key] = (waits, set(records), tails)
        return self._sync[key]

    def _lane_streams(self, key, n_lanes: int, device=None) -> list:
        pool = self._streams.setdefault(str(key), [])
        while len(pool) < n_lanes - 1:
            pool.append(torch.cuda.Stream(device=device if device is not None else key))
        return pool

    def _launch(self, lib, op: Op, bufs, B: int, stream) -> None:
        P = _lib.ptr
        if op.kind == "conv":
            op.desc.batch = B
            fn = lib.sp_conv3x3_direct if op.direct else lib.sp_conv2d_fwd
            _lib.check(fn(op.desc, P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift),
                          P(bufs[op.res]) if op.res else None, P(bufs[op.dst]), stream), op.name)
        elif op.kind in ("bb32", "bb64"):
            op.desc.batch = B
            w2, scale2, shift2 = op.args
            _lib.check((lib.sp_basic_block_c32 if op.kind == "bb32" else lib.sp_basic_block_c64)(op.desc, P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift), P(w2), P(scale2), P(shift2),
                                              P(bufs[op.dst]), stream), op.name)
        elif op.kind == "bneck64":
            op.desc.batch = B
            w1, s1, h1, w3, s3, h3 = op.args
            _lib.check(lib.sp_bottleneck_c64(op.desc, P(bufs[op.src]), P(w1), P(s1), P(h1), P(op.w), P(op.scale), P(op.shift), P(w3), P(s3), P(h3),
                                             P(bufs[op.dst]), stream), op.name)
        elif op.kind == "dual1x1":
            w_s, s_s, h_s, rows_per_image, relu = op.args
            fn = lib.sp_dual_pw_bf16 if op.w.element_size() == 2 else lib.sp_dual_pw_f32
            _lib.check(fn(P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift), P(bufs[op.res]), P(w_s), P(s_s), P(h_s), P(bufs[op.dst]),
                                           B * rows_per_image, 64, 64, 256, relu, stream), op.name)
        elif op.kind == "stem7":
            h, w, k_pad, unfused = op.args
            src = bufs[op.src]
            if src.dtype == torch.uint8:           # BGR crops [B,h,w,3]: normalised (coco.py:136) while the patch is loaded
                mean = (ctypes.c_float * 3)(0.485, 0.456, 0.406)
                _lib.check(lib.sp_stem7_pool_u8(P(src), mean, P(op.w), k_pad, P(op.scale), P(op.shift), P(bufs[op.dst]), int(self.dtype == "bf16"),
                                                B, h, w, stream), op.name)
            else:
                _lib.check(lib.sp_stem7_pool(P(src), P(op.w), k_pad, P(op.scale), P(op.shift), P(bufs[op.dst]), int(self.dtype == "bf16"),
                                             B, h, w, stream), op.name)
        elif op.kind == "htrans":
            h, w, k_pad, wb, sb, hb, _ = op.args
            _lib.check(lib.sp_hrnet_transition1(P(bufs[op.src]), B, h, w, P(op.w), k_pad, P(op.scale), P(op.shift), P(wb), P(sb), P(hb),
                                                P(bufs[op.dst]), P(bufs[op.dst2]), stream), op.name)
        elif op.kind == "hstem":
            h, w, k1_pad, w2, s2, h2, unfused = op.args
            src = bufs[op.src]
            if src.dtype == torch.uint8:           # BGR crops: the three-launch lowering, whose first launch normalises them
                tmp = dict(bufs)
                for u in unfused[:-1]:
                    uh, uw, uc = self.shapes[u.dst]
                    tmp[u.dst] = torch.empty(B * uh * uw * uc, dtype=torch.bfloat16, device=src.device)
                for u in unfused:
                    self._launch(lib, u, tmp, B, stream)
            else:
                _lib.check(lib.sp_hrnet_stem(P(src), P(op.w), k1_pad, P(op.scale), P(op.shift), P(w2), P(s2), P(h2), P(bufs[op.dst]), B, h, w,
                                             stream), op.name)
        elif op.kind == "maxpool":
            h, w, c = op.args
            fn = lib.sp_maxpool3x3s2_nhwc_bf16 if self.dtype == "bf16" else lib.sp_maxpool3x3s2_nhwc
            _lib.check(fn(P(bufs[op.src]), P(bufs[op.dst]), B, h, w, c, stream), op.name)
        elif op.kind == "to_nhwc4":
            c, h, w = op.args
            src = bufs[op.src]
            if src.dtype == torch.uint8:           # BGR crops [B,h,w,3]: normalise (coco.py:136) and lay out in one pass
                mean = (ctypes.c_float * 3)(0.485, 0.456, 0.406)
                _lib.check(lib.sp_u8hwc_bgr_to_nhwc(P(src), P(bufs[op.dst]), 2 if self.dtype == "bf16" else 0, B, h, w, mean, stream), op.name)
            else:
                fn = lib.sp_nchw_to_nhwc4_bf16 if self.dtype == "bf16" else lib.sp_nchw_to_nhwc4
                _lib.check(fn(P(src), P(bufs[op.dst]), B, c, h, w, stream), op.name)
        elif op.kind == "pixel_shuffle":
            h, w, c = op.args
            fn = lib.sp_pixel_shuffle2_nhwc_bf16 if self.dtype == "bf16" else lib.sp_pixel_shuffle2_nhwc
            _lib.check(fn(P(bufs[op.src]), P(bufs[op.dst]), B, h, w, c, stream), op.name)
        elif op.kind == "gap":
            hw, c = op.args
            fn = lib.sp_global_avg_pool_nhwc_bf16 if self.dtype == "bf16" else lib.sp_global_avg_pool_nhwc
            _lib.check(fn(P(bufs[op.src]), P(bufs[op.dst]), B, hw, c, stream), op.name)
        elif op.kind == "se_gate":
            hw, c, gate = op.args
            fn = lib.sp_se_gate_add_relu_nhwc_bf16 if self.dtype == "bf16" else lib.sp_se_gate_add_relu_nhwc
            _lib.check(fn(P(bufs[op.src]), P(bufs[gate]), P(bufs[op.res]), P(bufs[op.dst]), B, hw, c, stream), op.name)
        elif op.kind == "upsample_add_n":
            H, W, c, relu, more, factors = op.args
            srcs = (op.src,) + tuple(more)
            xs = (ctypes.c_void_p * len(srcs))(*[bufs[n].data_ptr() for n in srcs])
            fs = (ctypes.c_int32 * len(srcs))(*factors)
            _lib.check(lib.sp_upsample_add_n_nhwc(P(bufs[op.res]), int(self.dtype == "bf16"), len(srcs), xs, fs, P(bufs[op.dst]), B, H, W, c, relu,
                                                  stream), op.name)
        elif op.kind == "upsample_add":
            h, w, c, f, relu = op.args
            fn = lib.sp_upsample_add_nhwc_bf16 if self.dtype == "bf16" else lib.sp_upsample_add_nhwc
            _lib.check(fn(P(bufs[op.src]), P(bufs[op.res]), P(bufs[op.dst]), B, h, w, c, f, relu, stream), op.name)
        else:
            raise ValueError(op.kind)

    def run(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, slot: int = 0) -> torch.Tensor:
        """`out`: optional preallocated fp32 [B,J,H/4,W/4] result (a steady-state caller reuses one; default: a fresh tensor per call,
        which the caller may keep - torch's caching allocator makes that a pointer bump, no hipMalloc).
        x: fp32 NCHW [B,3,H,W] on the GPU (or uint8 BGR crops [B,H,W,3], normalised on the fly as datasets/coco.py:136 does) ->
        heat maps fp32 NCHW [B,J,H/4,W/4].  Ops are issued in program order; ops of
        different lanes (independent HRNet branches) go to different HIP streams and overlap on the GPU, ordered by events."""
        lib = _lib.lib()
        B = x.shape[0]
        bufs = dict(self._alloc(B, x.device, slot))       # (slot: engine.InterleavedForward - own activations, lane streams and events)
        bufs["input"] = x
        if out is None:
            out = torch.empty((B,) + tuple(self.out_shape), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (B,) + tuple(self.out_shape) or out.dtype != torch.float32 or out.device != x.device or not out.is_contiguous():
            raise ValueError(f"out: expected a contiguous fp32 {(B,) + tuple(self.out_shape)} tensor on {x.device}")
        bufs[self.out_name] = out
        if B == 0:                      # nothing to launch (a detector-driven caller with an image without persons)
            return out
        n_lanes = 1 + max((op.lane for op in self.ops), default=0) if self.multi_stream else 1
        if n_lanes == 1:
            stream = _lib.current_stream(x.device)             # the INPUT's device, not whatever device happens to be current
            for op in self.ops:
                self._launch(lib, op, bufs, B, stream)
            return out
        waits, records, tails = self._plan_sync(B, x.device, slot)
        main = torch.cuda.current_stream(x.device)
        side = self._lane_streams(x.device if slot == 0 else f"{x.device}/{slot}", n_lanes, x.device)
        streams = [main] + side[: n_lanes - 1]
        fork = torch.cuda.Event()
        fork.record(main)                                   # inputs are ready / the previous run has drained (it joined on `main`)
        for st in streams[1:]:
            st.wait_event(fork)
        handles = [ctypes.c_void_p(st.cuda_stream) for st in streams]
        events = self._events.setdefault(str(x.device) if slot == 0 else f"{x.device}/{slot}", {})
        for i, op in enumerate(self.ops):
            st = streams[op.lane]
            for j in waits[i]:
                st.wait_event(events[j])
            self._launch(lib, op, bufs, B, handles[op.lane])
            if i in records or (op.lane != 0 and tails[op.lane] == i):
                ev = events.get(i)
                if ev is None:
                    ev = events[i] = torch.cuda.Event()
                ev.record(st)
        for lane, i in tails.items():                       # join: the caller's stream owns the result and every buffer again
            if lane != 0:
                main.wait_event(events[i])
        return out                                          # (activation storage belongs to the program's pool: nothing to hand back)

    def capture(self, x: torch.Tensor, decoder=None, trans_inv: Optional[torch.Tensor] = None, warmup: int = 2) -> "GraphedForward":
        """Record the whole forward (and, when given, the key-point decode) of this batch shape into ONE hipGraph: the ~60 kernel
        launches of a step become a single graph launch (what matters at small batch, where the step is launch-bound)."""
        return GraphedForward(self, x, decoder, trans_inv, warmup)

    # -- per-layer tile autotuning -------------------------------------------------------------------------------
    def _candidates(self, lib, op: Op):
        """Every way this conv launch can run: (tile_m, tile_n, kernel) with kernel 0 = register-staged implicit GEMM, 1 = LDS-DMA ring
        (bf16 layers it supports), (-1, -1, 0) = the direct 3x3 kernel."""
        d = op.desc
        if d.c_in_group:                # grouped: the N tile IS the panel; implicit GEMM only
            return [(bm, bn, _lib.SP_CONV_KERNEL_IGEMM) for bm, bn in _lib.CONV_TILES if bn == d.c_in_group]
        keep = (d.tile_m, d.tile_n, d.kernel)
        out = [(bm, bn, _lib.SP_CONV_KERNEL_IGEMM) for bm, bn in _lib.CONV_TILES if d.n_pad % bn == 0]
        if d.flags & SP_CONV_BF16:
            for bm, bn in _lib.RING_TILES:
                d.tile_m, d.tile_n, d.kernel = bm, bn, _lib.SP_CONV_KERNEL_RING
                if lib.sp_conv2d_ring_ok(d):
                    out.append((bm, bn, _lib.SP_CONV_KERNEL_RING))
            for bm, bn in _lib.RING_LW_TILES:
                d.tile_m, d.tile_n, d.kernel = bm, bn, _lib.SP_CONV_KERNEL_RING_LW
                if lib.sp_conv2d_ring_ok(d):
                    out.append((bm, bn, _lib.SP_CONV_KERNEL_RING_LW))
            for bm, bn in _lib.RING_LW4_TILES:
                d.tile_m, d.tile_n, d.kernel = bm, bn, _lib.SP_CONV_KERNEL_RING_LW4
                if lib.sp_conv2d_ring_ok(d):
                    out.append((bm, bn, _lib.SP_CONV_KERNEL_RING_LW4))
        d.tile_m, d.tile_n, d.kernel = keep
        if lib.sp_conv2d_pw_ok(d):
            out.append((64, 256, _lib.SP_CONV_KERNEL_PW))
        if lib.sp_conv3x3_direct_ok(d):
            out.append((-1, -1, 0))
        return out

    def autotune(self, x: torch.Tensor, reps: int = 5, verbose: bool = False, rounds: int = 3,
                 refine: Optional[bool] = None, in_situ: bool = False) -> Dict[str, Tuple[int, int, int]]:
        """Time every legal (tile, kernel) of every distinct conv shape (HIP events on the launch stream, real activations of a
        warm-up pass as operands) and pin the fastest in the launch descriptors.  Results are bit-identical for every choice (same
        K reduction order), so this only moves speed.  `refine` (default: on for programs with several stream lanes): a second pass
        that re-decides the heaviest shapes on whole-step time (_refine_on_whole_step).  Per candidate: the MEDIAN of `rounds` timings of `reps` back-to-back
        launches; the two best are then re-timed interleaved with 3x the launches, so that a clock ramp or a noisy neighbour during
        one measurement does not decide the table (the driver saw one tile group 15 % slower than the builder's runs)."""
        lib = _lib.lib()
        B = x.shape[0]
        self.run(x)                                  # fills every activation buffer with realistic data
        bufs = dict(self._alloc(B, x.device))
        bufs["input"] = x
        bufs[self.out_name] = torch.empty((B,) + tuple(self.out_shape), dtype=torch.float32, device=x.device)
        stream = _lib.current_stream()
        P = _lib.ptr
        chosen: Dict[tuple, Tuple[int, int, int]] = {}
        report: Dict[str, Tuple[int, int, int]] = {}
        by_key: Dict[tuple, List[Op]] = {}
        ranked: Dict[tuple, list] = {}                 # per shape: [(isolated ms, candidate)] best first

        def apply(op, cand):
            d = op.desc
            op.direct = cand[0] < 0
            if not op.direct:
                d.tile_m, d.tile_n, d.kernel = cand

        def time_once(op, cand, n):
            d = op.desc
            apply(op, cand)
            fn = lib.sp_conv3x3_direct if cand[0] < 0 else lib.sp_conv2d_fwd
            args = (d, P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift), P(bufs[op.res]) if op.res else None, P(bufs[op.dst]), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(*args)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        for op in self.ops:
            if op.kind != "conv":
                continue
            d = op.desc
            d.batch = B
            key = tuple(getattr(d, f) for f, _ in ConvDesc._fields_ if f not in ("tile_m", "tile_n", "kernel")) + (op.res is not None,)
            if key not in chosen:
                cands = self._candidates(lib, op)
                timed = []
                for cand in cands:
                    apply(op, cand)
                    fn = lib.sp_conv3x3_direct if cand[0] < 0 else lib.sp_conv2d_fwd
                    _lib.check(fn(d, P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift), P(bufs[op.res]) if op.res else None,
                                  P(bufs[op.dst]), stream), op.name)               # warm-up (and validates the candidate)
                    ts = sorted(time_once(op, cand, reps) for _ in range(rounds))
                    # the 64-channel direct kernel owns a whole CU's LDS (one workgroup per CU): timed alone it ties with the implicit
                    # GEMM on HRNet's 32x24 branch, inside the multi-stream network it then keeps the other branches' workgroups off
                    # the CU (measured 26 vs 20 us per layer).  It has to win by a margin to be picked.
                    handicap = 1.10 if (cand[0] < 0 and d.c_in == 64) else 1.0
                    timed.append((ts[len(ts) // 2] * handicap, cand))
                    if verbose:
                        print(f"  {op.name:28s} {cand[0]:3d}x{cand[1]:<3d} k{cand[2]} {timed[-1][0] * 1e3:8.1f} us")
                timed.sort()
                ranked[key] = timed
                best = timed[0]
                if len(timed) > 1 and timed[1][0] < 1.08 * timed[0][0]:               # a close second: re-time both, interleaved
                    ta, tb = [], []
                    for _ in range(rounds):
                        ta.append(time_once(op, timed[0][1], 3 * reps))
                        tb.append(time_once(op, timed[1][1], 3 * reps))
                    ma, mb = sorted(ta)[rounds // 2], sorted(tb)[rounds // 2]
                    best = (ma, timed[0][1]) if ma <= mb else (mb, timed[1][1])
                chosen[key] = best[1]
            apply(op, chosen[key])
            report[op.name] = chosen[key]
            by_key.setdefault(key, []).append(op)
        self.tuned_for_batch = B
        n_lanes = 1 + max((op.lane for op in self.ops), default=0)
        if refine is None:
            refine = self.multi_stream and n_lanes > 1
        if in_situ and not (self.multi_stream and n_lanes > 1):     # (a multi-stream program is not what a one-stream forward times:
            report.update(self._retime_in_situ(x, by_key, ranked, chosen, apply, verbose))   #  there the whole-step refinement below decides)
        if refine:
            report.update(self._refine_on_whole_step(x, by_key, ranked, chosen, apply, verbose))
        return report

    def _retime_in_situ(self, x, by_key, ranked, chosen, apply, verbose, top: int = 4, reps: int = 3) -> Dict[str, Tuple[int, int, int]]:
        """Third opinion for the tile table: a launch repeated back to back rewrites the same output lines in the 256 MB Infinity Cache
        and re-reads hot operands, so candidates timed alone look 5-17 % faster than inside the network and do not always rank the same
        there.  Here the `top` candidates of every shape are timed where they run: the whole forward is executed on one stream with HIP
        events around the launches of that shape only (median of `reps` forwards per candidate), and the fastest in place is pinned."""
        lib = _lib.lib()
        B = x.shape[0]
        bufs = dict(self._alloc(B, x.device))
        bufs["input"] = x
        bufs[self.out_name] = torch.empty((B,) + tuple(self.out_shape), dtype=torch.float32, device=x.device)
        stream = _lib.current_stream(x.device)
        changed: Dict[str, Tuple[int, int, int]] = {}

        def forward_timing(ops_of_key) -> float:
            ids = {id(o) for o in ops_of_key}
            totals = []
            for _ in range(reps):
                evs = []
                for op in self.ops:
                    if id(op) in ids:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        self._launch(lib, op, bufs, B, stream)
                        e1.record()
                        evs.append((e0, e1))
                    else:
                        self._launch(lib, op, bufs, B, stream)
                torch.cuda.synchronize(x.device)
                totals.append(sum(a.elapsed_time(b) for a, b in evs))
            return sorted(totals)[len(totals) // 2]

        for key, ops_of_key in by_key.items():
            cands = [c for _, c in ranked[key][:top]]
            if len(cands) < 2:
                continue
            timed = []
            for cand in cands:
                for op in ops_of_key:
                    apply(op, cand)
                timed.append((forward_timing(ops_of_key), cand))
            best = min(timed)[1]
            if verbose and best != chosen[key]:
                print(f"  in situ {ops_of_key[0].name:28s} x{len(ops_of_key)}: {chosen[key]} -> {best}  " + " ".join(f"{c[0]}x{c[1]}k{c[2]}:{t * 1e3:.0f}us" for t, c in timed))
            chosen[key] = best
            for op in ops_of_key:
                apply(op, best)
                changed[op.name] = best
        return changed

    def _step_ms(self, x: torch.Tensor, steps: int = 5) -> float:
        """Milliseconds per whole run(x) (events on the caller's stream; run() joins every lane back onto it)."""
        for _ in range(2):
            self.run(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            self.run(x)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    def _refine_on_whole_step(self, x, by_key, ranked, chosen, apply, verbose) -> Dict[str, Tuple[int, int, int]]:
        """Second tuning pass for programs whose branches overlap on several streams (HRNet): a kernel timed ALONE is not the kernel
        that is best beside the other branches - the persistent ring / 64-channel direct kernels take a whole CU's LDS and keep other
        branches' workgroups off it, so their isolated wins can cost the step.  For every layer shape, heaviest first, the next-best
        candidates of the per-layer pass are tried in the whole step and kept only when the STEP gets faster (> 0.4 %)."""
        changed: Dict[str, Tuple[int, int, int]] = {}
        base = min(self._step_ms(x), self._step_ms(x))
        order = sorted(by_key, key=lambda k: -ranked[k][0][0] * len(by_key[k]))       # time share of the shape in the program
        for key in order[:10]:                             # the ten heaviest shapes carry > 90 % of the step
            alts = [c for _, c in ranked[key][1:3] if c != chosen[key]]
            # always offer the best candidate with a small LDS footprint (4-wave implicit GEMM) when the pick is a whole-CU kernel
            small = [c for _, c in ranked[key] if c[0] > 0 and c[2] == _lib.SP_CONV_KERNEL_IGEMM]
            if small and small[0] != chosen[key] and small[0] not in alts:
                alts.append(small[0])
            for cand in alts:
                for op in by_key[key]:
                    apply(op, cand)
                t = min(self._step_ms(x), self._step_ms(x))
                if t < base * 0.996:
                    if verbose:
                        print(f"  refine {by_key[key][0].name:28s} x{len(by_key[key])}: {chosen[key]} -> {cand}: step {base:.3f} -> {t:.3f} ms")
                    base, chosen[key] = t, cand
                else:
                    for op in by_key[key]:
                        apply(op, chosen[key])
            for op in by_key[key]:
                changed[op.name] = chosen[key]
        return changed

    def tiles(self) -> Dict[str, Tuple[int, int, int]]:
        return {op.name: ((-1, -1, 0) if op.direct else (op.desc.tile_m, op.desc.tile_n, op.desc.kernel)) for op in self.ops if op.kind == "conv"}

    def set_tiles(self, tiles: Dict[str, Tuple[int, ...]], batch: int) -> None:
        """Re-apply a tile table produced by autotune() (e.g. loaded from a file) instead of re-timing.  Entries are
        (tile_m, tile_n[, kernel]); (-1, -1) = the direct 3x3 kernel."""
        for op in self.ops:
            if op.kind == "conv" and op.name in tiles:
                t = [int(v) for v in tiles[op.name]]
                if op.desc.c_in_group:
                    # grouped conv: the N tile IS the weight panel and only the implicit GEMM reads block-diagonal panels.  Tables are
                    # keyed by layer name and resnet50 / resnext50 share names: an entry of another backbone must not reach this launch
                    if t[0] > 0 and t[1] == op.desc.c_in_group and (len(t) < 3 or t[2] == _lib.SP_CONV_KERNEL_IGEMM):
                        op.desc.tile_m = t[0]
                    continue
                op.direct = t[0] < 0
                if not op.direct:
                    op.desc.tile_m, op.desc.tile_n = t[0], t[1]
                    op.desc.kernel = t[2] if len(t) > 2 else _lib.SP_CONV_KERNEL_IGEMM
        self.tuned_for_batch = batch

    @property
    def flops_per_image(self) -> int:
        return sum(op.flops for op in self.ops)


class GraphedForward:
    """A Program (+ optional decoder) for one batch shape recorded as a hipGraph (torch.cuda.CUDAGraph is the recorder; every node
    is one of this library's kernels).  `static_input` / `static_trans_inv` are the graph's fixed input buffers: fill them in place
    (or pass tensors to __call__, which copies) and replay; the returned tensors are the graph's fixed outputs."""

    def __init__(self, prog: "Program", x: torch.Tensor, decoder=None, trans_inv: Optional[torch.Tensor] = None, warmup: int = 2):
        x = _lib.require_cuda_f32(x, "input")
        if decoder is not None and trans_inv is None:
            raise ValueError("capture with a decoder needs trans_inv")
        self.prog, self.decoder = prog, decoder
        self.static_input = x.clone()
        self.static_trans_inv = _lib.require_cuda_f32(trans_inv, "trans_inv").clone() if trans_inv is not None else None
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):                       # warm-up outside the capture: buffer pools get allocated here
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = self._body()
        # the graph's nodes hold raw device pointers into this batch shape's activation pool: keep the pool alive from here, whatever
        # Program._alloc evicts later (MAX_POOLS); a replay after an eviction would otherwise read and write freed memory
        self._pool = prog._pools[(x.shape[0], str(x.device))]

    def _body(self):
        # one stream inside the capture: hipStreamEndCapture of ROCm 7.2 crashed on the forked multi-stream HRNet schedule
        keep, self.prog.multi_stream = self.prog.multi_stream, False
        try:
            hm = self.prog.run(self.static_input)
        finally:
            self.prog.multi_stream = keep
        if self.decoder is None:
            return hm
        kps, mv = self.decoder(hm, self.static_trans_inv)
        return hm, kps, mv

    def __call__(self, x: Optional[torch.Tensor] = None, trans_inv: Optional[torch.Tensor] = None):
        if x is not None and x.data_ptr() != self.static_input.data_ptr():
            if x.shape != self.static_input.shape:
                raise ValueError(f"graph was captured for input {tuple(self.static_input.shape)}, got {tuple(x.shape)}")
            self.static_input.copy_(_lib.require_cuda_f32(x, "input"))
        if trans_inv is not None and self.static_trans_inv is not None and trans_inv.data_ptr() != self.static_trans_inv.data_ptr():
            self.static_trans_inv.copy_(_lib.require_cuda_f32(trans_inv, "trans_inv"))
        self.graph.replay()
        return self.outputs


class PipelinedForward:
    """forward + key-point decode as a two-stage pipeline over consecutive batches: the decode of batch i runs on its own HIP stream while the
    forward of batch i + 1 runs on the caller's (the decoder is VALU-bound - the reference's dense 11 x 11 blur replayed tap by tap - and the
    convolutions MFMA- / HBM-bound, so the two share the chip well; at bs = 128 the decode is 2 % of a bf16 step when it runs in line).
    Two heat-map buffers alternate; events order forward -> decode of the same batch and decode of batch i -> forward of batch i + 2.

        run = PipelinedForward(program, decoder)
        for x, tinv in batches: kps, score = run(x, tinv)      # valid once run.sync() (or a later stream-ordered read on run.stream)
    """

    def __init__(self, prog: "Program", decoder):
        self.prog, self.decoder = prog, decoder
        self.stream: Optional[torch.cuda.Stream] = None
        self._hm: list = [None, None]
        self._fwd_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._dec_done: list = [None, None]
        self._i = 0

    def __call__(self, x: torch.Tensor, trans_inv: torch.Tensor):
        dev = x.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        k = self._i & 1
        self._i += 1
        main = torch.cuda.current_stream(dev)
        shape = (x.shape[0],) + tuple(self.prog.out_shape)
        if self._hm[k] is None or tuple(self._hm[k].shape) != shape:
            self._hm[k] = torch.empty(shape, dtype=torch.float32, device=dev)
        if self._dec_done[k] is not None:
            main.wait_event(self._dec_done[k])            # the decode that last read this buffer
        hm = self.prog.run(x, out=self._hm[k])
        self._fwd_done[k].record(main)
        self.stream.wait_event(self._fwd_done[k])
        with torch.cuda.stream(self.stream):
            out = self.decoder(hm, trans_inv)
            if self._dec_done[k] is None:
                self._dec_done[k] = torch.cuda.Event()
            self._dec_done[k].record(self.stream)
        return out

    def sync(self) -> None:
        """The caller's stream waits for every decode issued so far."""
        for ev in self._dec_done:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)


class InterleavedForward:
    """Consecutive batches on `depth` independent streams, each with its own activation pool, lane streams and events: the forward (+ decode)
    of batch i + 1 starts while batch i is still in its low-occupancy tail.  For programs whose launches do not fill the chip (HRNet: 2-4
    branches of small convolutions, a last module that only produces one branch) this is where throughput is; the ResNets' launches fill
    the chip on their own.  Results are those of Program.run (+ decoder) bit for bit.

        run = InterleavedForward(program, decoder, depth=2)
        for x, tinv in batches: kps, score = run(x, tinv)      # valid after run.sync()
    """

    _next_slot = 1      # slots are unique per instance: two InterleavedForward objects on one Program never share activations

    def __init__(self, prog: "Program", decoder=None, depth: int = 2):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.prog, self.decoder, self.depth = prog, decoder, depth
        self._slot0 = InterleavedForward._next_slot
        InterleavedForward._next_slot += depth
        self._streams: list = []
        self._hm: list = [None] * depth
        self._done: list = [None] * depth
        self._ready = None
        self._i = 0

    def __call__(self, x: torch.Tensor, trans_inv: Optional[torch.Tensor] = None):
        dev = x.device
        while len(self._streams) < self.depth:
            self._streams.append(torch.cuda.Stream(device=dev))
        k = self._i % self.depth
        self._i += 1
        st = self._streams[k]
        if self._ready is None:
            self._ready = torch.cuda.Event()
        self._ready.record(torch.cuda.current_stream(dev))    # the inputs are ready on the caller's stream
        st.wait_event(self._ready)
        shape = (x.shape[0],) + tuple(self.prog.out_shape)
        if self._hm[k] is None or tuple(self._hm[k].shape) != shape:
            self._hm[k] = torch.empty(shape, dtype=torch.float32, device=dev)
        with torch.cuda.stream(st):
            out = self.prog.run(x, out=self._hm[k], slot=self._slot0 + k)
            if self.decoder is not None:
                out = self.decoder(out, trans_inv)
            if self._done[k] is None:
                self._done[k] = torch.cuda.Event()
            self._done[k].record(st)
        return out

    def sync(self) -> None:
        """The caller's stream waits for everything issued so far."""
        for ev in self._done:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def close(self) -> None:
        """Wait for the work in flight and give this object's activation pools, lane streams and events back."""
        torch.cuda.synchronize()
        mine = range(self._slot0, self._slot0 + self.depth)
        for key in [k for k in self.prog._pools if len(k) == 3 and k[2] in mine]:
            del self.prog._pools[key]
        for table in (self.prog._streams, self.prog._events):
            for key in [k for k in table if "/" in k and k.rsplit("/", 1)[1].isdigit() and int(k.rsplit("/", 1)[1]) in mine]:
                del table[key]
        self._hm = [None] * self.depth


class ProgramBuilder:
    """Appends launches to a Program while tracking NHWC buffer shapes."""

    def __init__(self, in_h: int, in_w: int, dtype: str = "fp32", packer=None):
        if dtype not in ("fp32", "bf16"):
            raise ValueError(dtype)
        self.packer = packer or HipPacker()         # parameters -> kernel layouts (device kernels behind the C ABI)
        # bf16: whole 32-channel BasicBlocks as one launch (sp_basic_block_c32: same bits, 2.5x less HBM traffic).  The round-2 four-wave kernel
        # tied with the two direct-conv launches it replaces (40 vs 2 x 20 us per block at bs=128, profiles/r02_pmc_hrnet_blocks.md); round 6's
        # eight-wave strip kernel takes 25.8 us against 2 x 16.5: hrnet_program turns this on, a bare builder leaves it to the caller
        self.fuse_blocks = False
        # ... and the 64-channel BasicBlocks (sp_basic_block_c64: conv1 waves / conv2 waves pipelined over 4-row strips).  Opt-in: bit-identical and 13.9 us
        # against 27.4 for the two launches at bs=32, but at bs=128 it ties with them alone (27.8 / 26.1 us) and costs HRNet-W32 0-4 % (profiles/r06_bb64_ab.txt)
        self.fuse_blocks64 = False
        # bf16: whole identity-shortcut Bottlenecks with 64 mid channels (ResNet-50 layer1.1 / layer1.2) as one launch
        # (sp_bottleneck_c64: same bits, x read once and y written once)
        self.fuse_bottlenecks = False
        # bf16: conv3 + the projection shortcut of a stage-opening Bottleneck with 64 mid channels (layer1.0 of the ResNets and of HRNet) as one
        # launch (sp_dual_pw_bf16: same bits, the 256-channel shortcut tensor is neither written nor read: 703 -> 301 MB at bs=128)
        self.fuse_tail = os.environ.get("SP_FUSE_TAIL", "1") != "0"       # (env: development knob for same-box A/Bs)
        # the ResNet stem (conv1 7x7 s2 + bn1 + relu + maxpool) as one launch on the fp32 NCHW image (sp_stem7_pool: same bits, the
        # 128 x 96 x 64 map between conv and pooling never reaches HBM, K is not padded to a GEMM tile)
        self.fuse_stem = True
        # HRNet fuse stage: the identity and upsampled terms of one output summed by ONE launch (sp_upsample_add_n_nhwc) instead of one
        # launch per term that re-reads and re-writes the running sum (43 -> 23 launches per HRNet-W32 forward; fp32: same bits)
        self.fuse_terms = True
        # bf16 HRNet: transition1's two 3x3 convolutions on layer1's 256-channel output (-> 32 channels at stride 1, -> 64 at stride 2) as
        # ONE launch (sp_hrnet_transition1: the halo staged once per 64-channel chunk serves both; 308 -> ~100 us and 1.4 GB -> 0.3 GB of
        # traffic at bs=128).  Reduction order (chunk, tap, channel): equal to the two conv launches up to fp32 summation order
        self.fuse_transition = True
        self.p = Program(dtype=dtype)
        self.bf16 = dtype == "bf16"
        self.cpad = 8 if self.bf16 else 4           # channels per 16-byte chunk
        self.p.shapes["input"] = (in_h, in_w, 3)
        self._n = 0
        self.lane = 0                               # ops appended from now on are issued on this lane (HIP stream)

    def _add(self, op: Op) -> None:
        op.lane = self.lane
        self.p.ops.append(op)

    def _fresh(self, stem: str) -> str:
        self._n += 1
        return f"{stem}#{self._n}"

    def to_nhwc4(self, src: str) -> str:
        h, w, c = self.p.shapes[src]
        dst = self._fresh("x4")
        # fp32: NHWC4 (one 16-byte chunk per pixel).  bf16: NHWC4 as well (8 bytes per pixel) - the stem then reads x-PAIRS of pixels
        # as 8-channel chunks (conv(): `paired`), which halves its K compared with padding every pixel to 8 channels
        self.p.shapes[dst] = (h, w, 4)
        self._add(Op("to_nhwc4", src, dst, args=(c, h, w), name="to_nhwc4"))
        return dst

    def stem_pool(self, src: str, weight: torch.Tensor, scale, shift, name: str = "conv1") -> str:
        """relu(bn(conv 7x7 s2 p3 (3 -> 64))) followed by maxpool 3x3 s2 p1 (pose_resnet_dconv.py:158-162).  With `fuse_stem` the three
        launches (layout change, implicit GEMM, pooling) become one `stem7` op that reads the fp32 NCHW image - or the uint8 BGR crops,
        normalising them on the way - directly; the three stay inside the op as its definition (the CPU interpreter of the test suite reads them)."""
        first = len(self.p.ops)
        x4 = self.to_nhwc4(src)
        y = self.conv(x4, weight, stride=2, pad=3, scale=scale, shift=shift, relu=True, name=name)
        out = self.maxpool(y)
        h, w, _ = self.p.shapes[src]
        if not (self.fuse_stem and tuple(weight.shape) == (64, 3, 7, 7) and _lib.lib().sp_stem7_pool_ok(1, h, w)):
            return out
        unfused = self.p.ops[first:]
        del self.p.ops[first:]
        conv = unfused[1]
        self._add(Op("stem7", src, out, w=conv.w, scale=scale, shift=shift, args=(h, w, conv.desc.k_pad, tuple(unfused)), name=name,
                     flops=conv.flops))
        return out

    def hrnet_stem(self, src: str, w1: torch.Tensor, s1, h1, w2: torch.Tensor, s2, h2) -> str:
        """relu(bn2(conv2(relu(bn1(conv1(x)))))), both 3x3 stride 2 (pose_hrnet.py:419-425).  bf16 with `fuse_stem`: one `hstem` op on the fp32
        NCHW image (sp_hrnet_stem; same bits); the three launches stay inside the op as its definition and for uint8 crop input."""
        first = len(self.p.ops)
        x4 = self.to_nhwc4(src)
        y = self.conv(x4, w1, stride=2, pad=1, scale=s1, shift=h1, relu=True, name="conv1")
        out = self.conv(y, w2, stride=2, pad=1, scale=s2, shift=h2, relu=True, name="conv2")
        h, w, _ = self.p.shapes[src]
        if not (self.bf16 and self.fuse_stem and tuple(w1.shape) == (64, 3, 3, 3) and tuple(w2.shape) == (64, 64, 3, 3)
                and _lib.lib().sp_hrnet_stem_ok(1, h, w)):
            return out
        unfused = self.p.ops[first:]
        del self.p.ops[first:]
        c1, c2 = unfused[1], unfused[2]
        self._add(Op("hstem", src, out, w=c1.w, scale=s1, shift=h1, args=(h, w, c1.desc.k_pad, c2.w, s2, h2, tuple(unfused)), name="stem",
                     flops=c1.flops + c2.flops))
        return out

    def maxpool(self, src: str) -> str:
        h, w, c = self.p.shapes[src]
        dst = self._fresh("pool")
        self.p.shapes[dst] = ((h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, c)
        self._add(Op("maxpool", src, dst, args=(h, w, c), name="maxpool"))
        return dst

    def conv(self, src: str, weight: torch.Tensor, *, stride: int = 1, pad: int = 0, scale=None, shift=None,
             relu: bool = False, res: Optional[str] = None, pixel_shuffle: bool = False, out_nchw: bool = False,
             dst: Optional[str] = None, name: str = "conv", groups: int = 1) -> str:
        h, w, c_buf = self.p.shapes[src]
        O, I, kh, kw = weight.shape
        paired = False
        pk = self.packer
        panel = 0
        if groups > 1:
            # grouped convolution (ResNeXt's conv2, pose_resnet_dconv.py:101): block-diagonal panels of `panel` channels, one per N tile; K per
            # tap is the panel, not c_in (sp_conv_desc.c_in_group).  64 = one bf16 K tile / two fp32 ones and a legal tile_n of the implicit GEMM.
            if not (O == c_buf and c_buf % groups == 0 and I == c_buf // groups and not pixel_shuffle and not out_nchw):
                raise NotImplementedError(f"{name}: grouped convolutions are lowered for c_out == c_in (got {tuple(weight.shape)} on {c_buf} channels)")
            panel = 64
            while panel % I:
                panel *= 2
            if O % panel or panel > 128:
                raise NotImplementedError(f"{name}: no panel width for {groups} groups of {I} channels in {O}")
            packed, th, tw, ci, k_pad = pk.grouped(weight, groups, panel, bf16=self.bf16), kh, kw, c_buf, kh * kw * panel
        elif self.bf16 and c_buf == 4 and I < 4:
            # bf16 stem on the NHWC4 image read as pixel pairs [h, w/2, 8]: pixel 2*ox - pad + kx = pair (ox - ceil(pad/2)) + pt, half
            # `sub`, with kx + s0 = 2*pt + sub.  A stride of 2 pixels is a stride of ONE pair: separate x / y strides (stride_x).
            if stride != 2 or w % 2:
                raise NotImplementedError("bf16 stem: stride-2 convolution on an even-width image expected")
            paired = True
            half = (pad + 1) // 2
            s0 = 2 * half - pad
            tpw = (kw - 1 + s0) // 2 + 1
            packed, th, tw, ci, k_pad = pk.conv(weight, c_in_pad=8, taps_w_pad=tpw, pair_s0=s0, bf16=True)
        elif c_buf == self.cpad and I < self.cpad:   # fp32 stem on NHWC4: pad channels to one chunk and the tap row to 8 / 4
            taps_w_pad = _round_up(kw, 8) if kw > 4 else 4
            packed, th, tw, ci, k_pad = pk.conv(weight, c_in_pad=self.cpad, taps_w_pad=taps_w_pad, bf16=self.bf16)
        else:
            assert I == c_buf, (name, I, c_buf)
            # pixel_shuffle: rows sub-pixel-major; `scale` / `shift` must come in the same order (fold_bn(..., pixel_shuffle=True))
            packed, th, tw, ci, k_pad = pk.conv(weight, pixel_shuffle=pixel_shuffle, bf16=self.bf16)
        gh, gw = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, ci
        d.grid_h, d.grid_w, d.c_out, d.n_pad = gh, gw, O, packed.shape[0]
        d.taps_h, d.taps_w, d.k_pad, d.stride = th, tw, k_pad, stride
        d.dy0, d.dy_step, d.dx0, d.dx_step = -pad, 1, -pad, 1
        if paired:
            d.in_w, d.stride_x, d.dx0 = w // 2, 1, -half
        d.phases_y = d.phases_x = 1
        flags = SP_CONV_RELU if relu else 0
        if pixel_shuffle:
            assert O % 4 == 0 and packed.shape[0] == O
            d.out_h, d.out_w, d.out_c = gh * 2, gw * 2, O // 4
            d.oy_mul = d.ox_mul = 2
            flags |= SP_CONV_PIXEL_SHUFFLE
        else:
            d.out_h, d.out_w, d.out_c = gh, gw, O
            d.oy_mul = d.ox_mul = 1
        d.oy_add = d.ox_add = 0
        if out_nchw:
            flags |= SP_CONV_OUT_NCHW
        if self.bf16:
            flags |= SP_CONV_BF16
        d.flags = flags
        if panel:
            d.c_in_group, d.tile_m, d.tile_n = panel, 128, panel
        dst = dst or self._fresh(name)
        self.p.shapes[dst] = (d.out_h, d.out_w, d.out_c)
        op = Op("conv", src, dst, res=res, desc=d, w=packed, scale=scale, shift=shift, name=name, flops=2 * gh * gw * O * I * kh * kw)
        # small-channel 3x3 layers (HRNet's 32-channel branch): the direct kernel is the default, the tuner may still pick a GEMM tile
        op.direct = bool(self.bf16 and _lib.lib().sp_conv3x3_direct_ok(d))
        self._add(op)
        return dst

    def basic_block_c32(self, src: str, w1: torch.Tensor, scale1, shift1, w2: torch.Tensor, scale2, shift2, name: str) -> Optional[str]:
        """HRNet BasicBlock on a 32- or 64-channel bf16 activation as one launch (sp_basic_block_c32 / sp_basic_block_c64); None when the shapes do
        not qualify."""
        h, w, c = self.p.shapes[src]
        if not (self.bf16 and self.fuse_blocks and c in (32, 64) and tuple(w1.shape) == (c, c, 3, 3) and tuple(w2.shape) == (c, c, 3, 3)):
            return None
        if c == 64 and not self.fuse_blocks64:
            return None
        p1, th, tw, ci, k_pad = self.packer.conv(w1, bf16=True)
        p2 = self.packer.conv(w2, bf16=True)[0]
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, ci
        d.grid_h, d.grid_w, d.c_out, d.n_pad = h, w, c, p1.shape[0]
        d.taps_h, d.taps_w, d.k_pad, d.stride = th, tw, k_pad, 1
        d.dy0, d.dy_step, d.dx0, d.dx_step = -1, 1, -1, 1
        d.phases_y = d.phases_x = 1
        d.out_h, d.out_w, d.out_c = h, w, c
        d.oy_mul = d.ox_mul = 1
        d.oy_add = d.ox_add = 0
        d.flags = SP_CONV_RELU | SP_CONV_BF16
        if not (_lib.lib().sp_basic_block_c32_ok(d) if c == 32 else _lib.lib().sp_basic_block_c64_ok(d)):
            return None
        dst = self._fresh(name)
        self.p.shapes[dst] = (h, w, c)
        self._add(Op("bb32" if c == 32 else "bb64", src, dst, desc=d, w=p1, scale=scale1, shift=shift1, args=(p2, scale2, shift2), name=name,
                     flops=2 * (2 * h * w * c * c * 9)))
        return dst

    def bottleneck_c64(self, src: str, w1, s1, h1, w2, s2, h2, w3, s3, h3, name: str) -> Optional[str]:
        """Identity-shortcut Bottleneck 256 -> 64 -> 64 -> 256 on a bf16 activation as one launch (sp_bottleneck_c64); None when the shapes
        do not qualify or the fusion is off."""
        h, w, c = self.p.shapes[src]
        if not (self.bf16 and self.fuse_bottlenecks and c == 256 and tuple(w1.shape) == (64, 256, 1, 1) and tuple(w2.shape) == (64, 64, 3, 3)
                and tuple(w3.shape) == (256, 64, 1, 1)):
            return None
        p1 = self.packer.conv(w1, bf16=True)[0]
        p2, th, tw, ci, k_pad = self.packer.conv(w2, bf16=True)
        p3 = self.packer.conv(w3, bf16=True)[0]
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, ci
        d.grid_h, d.grid_w, d.c_out, d.n_pad = h, w, 64, p2.shape[0]
        d.taps_h, d.taps_w, d.k_pad, d.stride = th, tw, k_pad, 1
        d.dy0, d.dy_step, d.dx0, d.dx_step = -1, 1, -1, 1
        d.phases_y = d.phases_x = 1
        d.out_h, d.out_w, d.out_c = h, w, 64
        d.oy_mul = d.ox_mul = 1
        d.oy_add = d.ox_add = 0
        d.flags = SP_CONV_RELU | SP_CONV_BF16
        if not _lib.lib().sp_bottleneck_c64_ok(d) or p1.shape[1] != 256 or p3.shape != (256, 64):
            return None
        dst = self._fresh(name)
        self.p.shapes[dst] = (h, w, 256)
        self._add(Op("bneck64", src, dst, desc=d, w=p2, scale=s2, shift=h2, args=(p1, s1, h1, p3, s3, h3), name=name,
                     flops=2 * h * w * (256 * 64 + 64 * 64 * 9 + 64 * 256)))
        return dst

    def dual_pointwise_tail(self, main: str, w3, s3, h3, short: str, wd, sd_, hd, name: str) -> Optional[str]:
        """conv3 + the projection shortcut of a stage-opening Bottleneck with 64 mid channels as one launch (sp_dual_pw_bf16 / sp_dual_pw_f32): y =
        relu(bn3(conv3(main)) + bn_d(conv_d(short))); None when the shapes do not qualify or the fusion is off."""
        h, w, c = self.p.shapes[main]
        hs, ws_, cs = self.p.shapes[short]
        ok = _lib.lib().sp_dual_pw_bf16_ok if self.bf16 else _lib.lib().sp_dual_pw_f32_ok
        if not (self.fuse_tail and (h, w) == (hs, ws_) and c == 64 and cs == 64 and tuple(w3.shape) == (256, 64, 1, 1)
                and tuple(wd.shape) == (256, 64, 1, 1) and ok(h * w, 64, 64, 256)):
            return None
        p3, pd = self.packer.conv(w3, bf16=self.bf16)[0], self.packer.conv(wd, bf16=self.bf16)[0]
        if tuple(p3.shape) != (256, 64) or tuple(pd.shape) != (256, 64):
            return None
        dst = self._fresh(name)
        self.p.shapes[dst] = (h, w, 256)
        self._add(Op("dual1x1", main, dst, res=short, w=p3, scale=s3, shift=h3, args=(pd, sd_, hd, h * w, 1), name=name, flops=2 * h * w * 2 * 64 * 256))
        return dst

    def deconv_k4s2p1(self, src: str, weight: torch.Tensor, *, scale=None, shift=None, relu: bool = False,
                      name: str = "deconv") -> str:
        h, w, c = self.p.shapes[src]
        I, O = weight.shape[:2]
        assert I == c
        packed, n_pad = self.packer.deconv(weight, bf16=self.bf16)
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, I
        d.grid_h, d.grid_w, d.c_out, d.n_pad = h, w, O, n_pad
        d.taps_h, d.taps_w, d.k_pad, d.stride = 2, 2, 4 * I, 1
        d.dy0, d.dy_step, d.dx0, d.dx_step = 0, -1, 0, -1       # + phase (py, px) inside the kernel
        d.out_h, d.out_w, d.out_c = 2 * h, 2 * w, O
        d.oy_mul, d.oy_add, d.ox_mul, d.ox_add = 2, 0, 2, 0    # + phase
        d.phases_y = d.phases_x = 2
        d.flags = (SP_CONV_RELU if relu else 0) | (SP_CONV_BF16 if self.bf16 else 0)
        dst = self._fresh(name)
        self.p.shapes[dst] = (2 * h, 2 * w, O)
        self._add(Op("conv", src, dst, desc=d, w=packed, scale=scale, shift=shift,
                             name=name, flops=2 * h * w * I * O * 16))
        return dst

    def pixel_shuffle(self, src: str) -> str:
        h, w, c = self.p.shapes[src]
        dst = self._fresh("pshuf")
        self.p.shapes[dst] = (2 * h, 2 * w, c // 4)
        self._add(Op("pixel_shuffle", src, dst, args=(h, w, c), name="pixel_shuffle"))
        return dst

    def gap(self, src: str) -> str:
        h, w, c = self.p.shapes[src]
        dst = self._fresh("gap")
        self.p.shapes[dst] = (1, 1, c)
        self._add(Op("gap", src, dst, args=(h * w, c), name="se.avg_pool"))
        return dst

    def se_gate(self, x: str, gate_logits: str, identity: str) -> str:
        h, w, c = self.p.shapes[x]
        dst = self._fresh("se")
        self.p.shapes[dst] = (h, w, c)
        self._add(Op("se_gate", x, dst, res=identity, args=(h * w, c, gate_logits), name="se.gate_add_relu"))
        return dst

    def upsample_add(self, src: str, base: str, factor: int, relu: bool = False) -> str:
        """dst = base + nearest_upsample(src, factor) (+ relu); factor 1 = plain add."""
        h, w, c = self.p.shapes[src]
        assert self.p.shapes[base] == (h * factor, w * factor, c), (self.p.shapes[base], (h, w, c), factor)
        dst = self._fresh("fuse")
        self.p.shapes[dst] = self.p.shapes[base]
        self._add(Op("upsample_add", src, dst, res=base, args=(h, w, c, factor, int(relu)), name="upsample_add"))
        return dst


    def upsample_add_n(self, base: str, terms: List[Tuple[str, int]], relu: bool = False) -> str:
        """dst = [relu](((base + up(t0, f0)) + up(t1, f1)) + up(t2, f2)): every term of an HRNet fuse output that is added AFTER the
        stride-2 chains, in ONE launch (sp_upsample_add_n_nhwc; factor 1 = the identity term).  1..3 terms."""
        assert 1 <= len(terms) <= 3
        H, W, c = self.p.shapes[base]
        for src, f in terms:
            h, w, cc = self.p.shapes[src]
            assert (h * f, w * f, cc) == (H, W, c), (self.p.shapes[src], f, (H, W, c))
        dst = self._fresh("fuse")
        self.p.shapes[dst] = (H, W, c)
        self._add(Op("upsample_add_n", terms[0][0], dst, res=base,
                     args=(H, W, c, int(relu), tuple(t for t, _ in terms[1:]), tuple(f for _, f in terms)), name="upsample_add_n"))
        return dst


# ------------------------------------------------------------------------------------------------
# ResNet-50 (+ DConv / DUC head)
# ------------------------------------------------------------------------------------------------
def _bn(b: "ProgramBuilder", sd, prefix, pixel_shuffle: bool = False):
    return b.packer.fold_bn(sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            pixel_shuffle=pixel_shuffle)


def _bottleneck(b: ProgramBuilder, sd, x: str, p: str, stride: int) -> str:
    if b.fuse_bottlenecks and stride == 1 and (p + ".downsample.0.weight") not in sd and (p + ".se.fc.0.weight") not in sd:
        s1, h1 = _bn(b, sd, p + ".bn1")
        s2, h2 = _bn(b, sd, p + ".bn2")
        s3, h3 = _bn(b, sd, p + ".bn3")
        y = b.bottleneck_c64(x, sd[p + ".conv1.weight"], s1, h1, sd[p + ".conv2.weight"], s2, h2, sd[p + ".conv3.weight"], s3, h3, name=p)
        if y is not None:
            return y
    s1, h1 = _bn(b, sd, p + ".bn1")
    t = b.conv(x, sd[p + ".conv1.weight"], scale=s1, shift=h1, relu=True, name=p + ".conv1")
    s2, h2 = _bn(b, sd, p + ".bn2")
    w2 = sd[p + ".conv2.weight"]
    t = b.conv(t, w2, stride=stride, pad=1, scale=s2, shift=h2, relu=True, name=p + ".conv2", groups=w2.shape[0] // w2.shape[1])   # (resnext*: groups = 32)
    idn = x
    if (p + ".downsample.0.weight") in sd and stride == 1 and (p + ".se.fc.0.weight") not in sd and w2.shape[0] == w2.shape[1]:
        # layer1.0 (64 mid channels): conv3 + the projection shortcut as one launch - the 256-channel shortcut tensor is never written (same bits)
        sdn, hdn = _bn(b, sd, p + ".downsample.1")
        s3, h3 = _bn(b, sd, p + ".bn3")
        y = b.dual_pointwise_tail(t, sd[p + ".conv3.weight"], s3, h3, x, sd[p + ".downsample.0.weight"], sdn, hdn, name=p + ".conv3+downsample")
        if y is not None:
            return y
    if (p + ".downsample.0.weight") in sd:
        sdn, hdn = _bn(b, sd, p + ".downsample.1")
        # (stays on the main path's stream: giving the shortcut its own stream measured -1 % fp32 / -4 % bf16 at bs=128 - these
        # launches fill the chip on their own, unlike HRNet's low-resolution branches)
        idn = b.conv(x, sd[p + ".downsample.0.weight"], stride=stride, scale=sdn, shift=hdn, name=p + ".downsample")
    s3, h3 = _bn(b, sd, p + ".bn3")
    if (p + ".se.fc.0.weight") in sd:
        # SE variant (reduction=True): out = relu(se(bn3(conv3(t))) + identity), pose_resnet_dconv.py:124-131
        z = b.conv(t, sd[p + ".conv3.weight"], scale=s3, shift=h3, name=p + ".conv3")
        g = b.gap(z)
        g = b.conv(g, sd[p + ".se.fc.0.weight"], shift=b.packer.bias(sd[p + ".se.fc.0.bias"]), relu=True, name=p + ".se.fc.0")
        g = b.conv(g, sd[p + ".se.fc.2.weight"], shift=b.packer.bias(sd[p + ".se.fc.2.bias"]), name=p + ".se.fc.2")
        return b.se_gate(z, g, idn)
    # bn3 + residual add + relu fused in conv3's epilogue (pose_resnet_dconv.py:124-131)
    return b.conv(t, sd[p + ".conv3.weight"], scale=s3, shift=h3, relu=True, res=idn, name=p + ".conv3")


def _res_basic_block(b: ProgramBuilder, sd, x: str, p: str, stride: int) -> str:
    """BasicBlock.forward of resnet18 / resnet34 (pose_resnet_dconv.py:61-80): conv3x3 (stride) - bn - relu, conv3x3 - bn [- SELayer],
    + identity (or the 1x1 projection shortcut), relu."""
    s1, h1 = _bn(b, sd, p + ".bn1")
    t = b.conv(x, sd[p + ".conv1.weight"], stride=stride, pad=1, scale=s1, shift=h1, relu=True, name=p + ".conv1")
    idn = x
    if (p + ".downsample.0.weight") in sd and stride == 1 and (p + ".se.fc.0.weight") not in sd and w2.shape[0] == w2.shape[1]:
        # layer1.0 (64 mid channels): conv3 + the projection shortcut as one launch - the 256-channel shortcut tensor is never written (same bits)
        sdn, hdn = _bn(b, sd, p + ".downsample.1")
        s3, h3 = _bn(b, sd, p + ".bn3")
        y = b.dual_pointwise_tail(t, sd[p + ".conv3.weight"], s3, h3, x, sd[p + ".downsample.0.weight"], sdn, hdn, name=p + ".conv3+downsample")
        if y is not None:
            return y
    if (p + ".downsample.0.weight") in sd:
        sdn, hdn = _bn(b, sd, p + ".downsample.1")
        idn = b.conv(x, sd[p + ".downsample.0.weight"], stride=stride, scale=sdn, shift=hdn, name=p + ".downsample")
    s2, h2 = _bn(b, sd, p + ".bn2")
    if (p + ".se.fc.0.weight") in sd:
        z = b.conv(t, sd[p + ".conv2.weight"], pad=1, scale=s2, shift=h2, name=p + ".conv2")
        g = b.gap(z)
        g = b.conv(g, sd[p + ".se.fc.0.weight"], shift=b.packer.bias(sd[p + ".se.fc.0.bias"]), relu=True, name=p + ".se.fc.0")
        g = b.conv(g, sd[p + ".se.fc.2.weight"], shift=b.packer.bias(sd[p + ".se.fc.2.bias"]), name=p + ".se.fc.2")
        return b.se_gate(z, g, idn)
    return b.conv(t, sd[p + ".conv2.weight"], pad=1, scale=s2, shift=h2, relu=True, res=idn, name=p + ".conv2")


def resnet_program(sd: Dict[str, torch.Tensor], head: str, in_h: int = 256, in_w: int = 192,
                   blocks=(3, 4, 6, 3), dtype: str = "fp32", packer=None, fuse_bottlenecks: bool = False, fuse_stem: bool = True) -> Program:
    """Lower a reference-layout state_dict into a Program.  `sd` tensors must be on the GPU.
    `fuse_bottlenecks`: bf16 identity-shortcut Bottlenecks with 64 mid channels as one launch each (sp_bottleneck_c64; same bits)."""
    b = ProgramBuilder(in_h, in_w, dtype, packer)
    b.fuse_bottlenecks = fuse_bottlenecks
    b.fuse_tail = fuse_bottlenecks and b.fuse_tail      # (one switch for the Bottleneck fusions of the ResNet programs: the per-conv program is the bitwise reference)
    b.fuse_stem = fuse_stem
    s, h = _bn(b, sd, "bn1")
    x = b.stem_pool("input", sd["conv1.weight"], s, h, name="conv1")
    basic = "layer1.0.conv3.weight" not in sd            # resnet18 / resnet34: BasicBlocks (two 3x3 convs, no conv3)
    for li, n in enumerate(blocks, start=1):
        for bi in range(n):
            x = (_res_basic_block if basic else _bottleneck)(b, sd, x, f"layer{li}.{bi}", 2 if (bi == 0 and li > 1) else 1)
    J = sd["final_layer.weight"].shape[0]
    if head == "dconv":
        for idx in (0, 3, 6):
            s, h = _bn(b, sd, f"deconv_layers.{idx + 1}")
            x = b.deconv_k4s2p1(x, sd[f"deconv_layers.{idx}.weight"], scale=s, shift=h, relu=True,
                                name=f"deconv_layers.{idx}")
        b.conv(x, sd["final_layer.weight"], shift=b.packer.bias(sd["final_layer.bias"]), out_nchw=True,
               dst="heat", name="final_layer")
    elif head == "duc":
        x = b.pixel_shuffle(x)
        for idx in (1, 2):
            s, h = _bn(b, sd, f"duc_layers.{idx}.bn", pixel_shuffle=True)    # in the packed (sub-pixel-major) column order
            x = b.conv(x, sd[f"duc_layers.{idx}.conv.weight"], pad=1, scale=s, shift=h, relu=True, pixel_shuffle=True,
                       name=f"duc_layers.{idx}")
        b.conv(x, sd["final_layer.weight"], pad=1, shift=b.packer.bias(sd["final_layer.bias"]), out_nchw=True,
               dst="heat", name="final_layer")
    else:
        raise ValueError(head)
    hh, ww, _ = b.p.shapes["heat"]
    b.p.out_shape = (J, hh, ww)
    return b.p


# ------------------------------------------------------------------------------------------------
# HRNet (nets/pose_hrnet.py)
# ------------------------------------------------------------------------------------------------
def _basic_block(b: ProgramBuilder, sd, x: str, p: str) -> str:
    """BasicBlock.forward (pose_hrnet.py:34-51): conv3x3-bn-relu, conv3x3-bn, + x, relu (stride 1, no downsample)."""
    s1, h1 = _bn(b, sd, p + ".bn1")
    s2, h2 = _bn(b, sd, p + ".bn2")
    fused = b.basic_block_c32(x, sd[p + ".conv1.weight"], s1, h1, sd[p + ".conv2.weight"], s2, h2, name=p)
    if fused is not None:
        return fused
    t = b.conv(x, sd[p + ".conv1.weight"], pad=1, scale=s1, shift=h1, relu=True, name=p + ".conv1")
    return b.conv(t, sd[p + ".conv2.weight"], pad=1, scale=s2, shift=h2, relu=True, res=x, name=p + ".conv2")


def _hr_module(b: ProgramBuilder, sd, xs: List[str], base: str, num_blocks: List[int], multi: bool) -> List[str]:
    """HighResolutionModule.forward (pose_hrnet.py:241-259)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        b.lane = i                                         # the branches of a module are independent: one HIP stream each
        for k in range(num_blocks[i]):
            xs[i] = _basic_block(b, sd, xs[i], f"{base}.branches.{i}.{k}")
    outs = []
    for i in range(nb if multi else 1):
        b.lane = i                                         # output i of the fuse stage is assembled on branch i's stream
        y: Optional[str] = None
        pending: List[Tuple[str, int]] = []                # terms j >= i: summed in ONE launch once the last one exists (fuse_terms)
        for j in range(nb):
            last = (j == nb - 1)
            f = f"{base}.fuse_layers.{i}.{j}"
            if j == i:
                if y is None:
                    y = xs[i]
                    if last:       # single term: cannot happen (nb >= 2), kept for completeness
                        raise NotImplementedError
                elif b.fuse_terms:
                    pending.append((xs[i], 1))
                else:
                    y = b.upsample_add(xs[i], y, 1, relu=last)
            elif j > i:            # 1x1 conv + bn at the low resolution, nearest upsample, add (:192-202)
                s, h = _bn(b, sd, f + ".1")
                t = b.conv(xs[j], sd[f + ".0.weight"], scale=s, shift=h, name=f)
                if b.fuse_terms:
                    pending.append((t, 2 ** (j - i)))
                else:
                    y = b.upsample_add(t, y, 2 ** (j - i), relu=last)
            if last and pending:   # (j < i never is the last j: the identity term j == i follows it)
                y = b.upsample_add_n(y, pending, relu=True)
                pending = []
            if j >= i:
                continue
            else:                  # chain of 3x3 stride-2 convs (+bn, +relu except the last) (:205-233)
                t = xs[j]
                for k in range(i - j):
                    s, h = _bn(b, sd, f"{f}.{k}.1")
                    fin = (k == i - j - 1)
                    t = b.conv(t, sd[f"{f}.{k}.0.weight"], stride=2, pad=1, scale=s, shift=h,
                               relu=(last if fin else True), res=(y if fin else None), name=f"{f}.{k}")
                y = t
        outs.append(y)
    b.lane = 0
    return outs


def _fuse_transition1(b: "ProgramBuilder") -> None:
    """The last two ops, if they are transition1's pair - conv3x3(256 -> 32, s1) and conv3x3(256 -> 64, s2) of the same input, BN + ReLU,
    bf16 - become one `htrans` op (sp_hrnet_transition1).  The two conv ops stay inside it as its definition."""
    if not (b.bf16 and b.fuse_transition and len(b.p.ops) >= 2):
        return
    a, c = b.p.ops[-2], b.p.ops[-1]
    if not (a.kind == "conv" and c.kind == "conv" and a.src == c.src and a.res is None and c.res is None):
        return
    da, dc = a.desc, c.desc
    ok = (da.c_in == 256 and dc.c_in == 256 and da.taps_h == 3 and da.taps_w == 3 and dc.taps_h == 3 and dc.taps_w == 3 and da.stride == 1 and
          dc.stride == 2 and da.c_out == 32 and dc.c_out == 64 and da.n_pad == 32 and dc.n_pad == 64 and da.k_pad == 2304 and dc.k_pad == 2304 and
          (da.flags & SP_CONV_RELU) and (dc.flags & SP_CONV_RELU) and da.dy0 == -1 and dc.dy0 == -1 and
          a.scale is not None and c.scale is not None and _lib.lib().sp_hrnet_transition1_ok(256, da.in_h, da.in_w))
    if not ok:
        return
    del b.p.ops[-2:]
    b._add(Op("htrans", a.src, a.dst, dst2=c.dst, w=a.w, scale=a.scale, shift=a.shift, args=(da.in_h, da.in_w, da.k_pad, c.w, c.scale, c.shift, (a, c)),
              name="transition1", flops=a.flops + c.flops))


def hrnet_program(sd: Dict[str, torch.Tensor], cfg: dict, in_h: int = 256, in_w: int = 192, dtype: str = "fp32", packer=None,
                  fuse_blocks: bool = True, fuse_blocks64: bool = False, fuse_stem: bool = True, fuse_terms: bool = True, fuse_transition: bool = True, fuse_tail: bool = True,
                  fuse_bottlenecks: bool = True) -> Program:
    """Lower a reference-layout HRNet state_dict into a Program (PoseHighResolutionNet.forward, pose_hrnet.py:419-454).
    `fuse_blocks`: bf16 32-channel BasicBlocks as one launch each (sp_basic_block_c32; same bits as the two conv launches; default since
    round 6's eight-wave strip kernel: 25.8 against 2 x 16.5 us per block at bs=128, HRNet-W32 +6.5 %)."""
    extra = cfg["MODEL"]["EXTRA"]
    b = ProgramBuilder(in_h, in_w, dtype, packer)
    b.fuse_blocks = {"1": True, "0": False}.get(os.environ.get("SP_HRNET_BLOCKS", ""), fuse_blocks)     # (env: development knob for same-box A/Bs)
    b.fuse_blocks64 = {"1": True, "0": False}.get(os.environ.get("SP_HRNET_BLOCKS64", ""), fuse_blocks64)
    b.fuse_tail = fuse_tail and b.fuse_tail
    # layer1.1-1.3 (identity Bottlenecks 256 -> 64 -> 64 -> 256) through sp_bottleneck_c64: neutral with the 171 / 168 us kernels of rounds 4-6, +2.3 % with
    # the eight-wave kernel (146 us against ~190 for the three launches; profiles/r06_summary.md)
    b.fuse_bottlenecks = fuse_bottlenecks and os.environ.get("SP_HRNET_BNECK", "1") != "0"
    b.fuse_terms = fuse_terms
    b.fuse_transition = fuse_transition
    b.fuse_stem = fuse_stem
    s1, h1 = _bn(b, sd, "bn1")
    s2, h2 = _bn(b, sd, "bn2")
    x = b.hrnet_stem("input", sd["conv1.weight"], s1, h1, sd["conv2.weight"], s2, h2)
    for k in range(4):
        x = _bottleneck(b, sd, x, f"layer1.{k}", 1)
    ys = [x]
    pre_n = 1
    for si, st in enumerate((2, 3, 4)):
        sc = extra[f"STAGE{st}"]
        nb = sc["NUM_BRANCHES"]
        t = f"transition{si + 1}"
        xs: List[str] = []
        for i in range(nb):                              # :427-432 / :436-441 / :445-450
            b.lane = i
            if i < pre_n:
                if (f"{t}.{i}.0.weight") in sd:
                    s_, h_ = _bn(b, sd, f"{t}.{i}.1")
                    xs.append(b.conv(ys[i], sd[f"{t}.{i}.0.weight"], pad=1, scale=s_, shift=h_, relu=True, name=f"{t}.{i}"))
                else:
                    xs.append(ys[i])
            else:
                v = ys[-1]
                for j in range(i + 1 - pre_n):
                    s_, h_ = _bn(b, sd, f"{t}.{i}.{j}.1")
                    v = b.conv(v, sd[f"{t}.{i}.{j}.0.weight"], stride=2, pad=1, scale=s_, shift=h_, relu=True, name=f"{t}.{i}.{j}")
                xs.append(v)
        b.lane = 0
        if si == 0:
            _fuse_transition1(b)
        for m in range(sc["NUM_MODULES"]):
            multi = not (st == 4 and m == sc["NUM_MODULES"] - 1)
            xs = _hr_module(b, sd, xs, f"stage{st}.{m}", list(sc["NUM_BLOCKS"]), multi)
        ys = xs
        pre_n = nb
    kf = extra["FINAL_CONV_KERNEL"]
    b.conv(ys[0], sd["final_layer.weight"], pad=1 if kf == 3 else 0, shift=b.packer.bias(sd["final_layer.bias"]),
           out_nchw=True, dst="heat", name="final_layer")
    hh, ww, _ = b.p.shapes["heat"]
    b.p.out_shape = (sd["final_layer.weight"].shape[0], hh, ww)
    return b.p
