"""Training step of the ResNet-50 DConv pose net on libsimple_pose_hip.so (fp32).

Replaces the body of `DDPProcessor.train` (processors/ddp_pose_resnet_solver.py:110-133):

    predicts = model(x)                                   -> train-mode forward: conv (MFMA) + batch-stat BN kernels
    loss = 0.5 * MSE(predicts * mask, targets * mask)     -> sp_masked_mse (value + d loss / d predicts)
    loss.backward()                                       -> dgrad (the forward conv kernel on re-packed weights),
                                                             wgrad (conv_wgrad kernel), BN / ReLU / max-pool backward
    DDP gradient all-reduce (:91-93)                      -> ONE all-reduce of the flat gradient buffer (RCCL)
    optimizer.step()   (Adam, :70-72)                     -> sp_adam_step over the flat parameter buffer

Parameters stay `nn.Parameter`s of the reference-layout module (views into one flat fp32 buffer, `.grad` views into a
flat gradient buffer), so `state_dict()`, checkpoints and any torch optimizer keep working; the packed weight copies
the kernels read are regenerated on the device after every update (sp_permute4_f32).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, SP_CONV_BF16, SP_CONV_BN_Y_MASK, SP_CONV_OUT_F32, SP_CONV_OUT_NCHW, SP_CONV_RELU
from .engine import _round_up, n_pad_for

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
P = _lib.ptr


def _i32(*v):
    return (ctypes.c_int32 * len(v))(*v)


def _i64(*v):
    return (ctypes.c_int64 * len(v))(*v)


def flat_layout(named_shapes) -> Tuple[Dict[str, Tuple[int, int]], int]:
    """Offsets of the parameters in the flat buffers (FlatParams' rule: parameter order, every view 16-byte aligned) from (name, shape)
    pairs alone: (offsets {name: (offset, numel)}, total floats)."""
    offsets: Dict[str, Tuple[int, int]] = {}
    off = 0
    for n, shape in named_shapes:
        k = 1
        for d in shape:
            k *= int(d)
        offsets[n] = (off, k)
        off += _round_up(k, 4)
    return offsets, _round_up(off, 4)


def plan_gradient_buckets(offsets: Dict[str, Tuple[int, int]], numel: int, bucket_mb: float, stem_params: int = 3) -> List[dict]:
    """Contiguous slices of the flat gradient buffer, cut at parameter boundaries walking from the END of the buffer (backward produces
    final_layer first); each bucket knows which parameter gradients it waits for.  Pure host arithmetic (no device): PoseTrainer uses it
    at construction, `bench.py --dry-launch` to print the plan of a world-8 job on the CPU.

    `stem_params` (round 5): the first parameters of the net (conv1.weight, bn1.weight, bn1.bias - the stem, whose gradients are the LAST to
    exist) get a bucket of their own.  With the optimizer inside backward a bucket's all-reduce -> Adam -> repack starts when its last
    gradient lands; in one bucket with layer1 those ~24 MB of optimizer work sat exposed behind the stem's backward and its weight gradient
    (the one-step timeline of round 4: 220 us from the chain's last kernel to the end of the step).  Now layer1's bucket goes out when
    layer1.0 is done, under the stem's backward, and what follows the chain is the stem's own 9,536 parameters."""
    names = list(offsets.keys())
    cap = max(1, int(bucket_mb * (1 << 20) / 4))
    stem = set(names[:stem_params]) if 0 < stem_params < len(names) else set()
    buckets: List[dict] = []
    hi = numel
    cur: List[str] = []
    lo = hi
    for n in reversed(names):
        if stem and n in stem and cur and not (set(cur) & stem):
            buckets.append({"lo": lo, "hi": hi, "names": set(cur)})       # everything behind the stem closes here, whatever its size
            hi, cur = lo, []
        o, _ = offsets[n]
        cur.append(n)
        lo = o
        if hi - lo >= cap:
            buckets.append({"lo": lo, "hi": hi, "names": set(cur)})
            hi, cur = lo, []
    if cur or hi > 0:
        buckets.append({"lo": 0, "hi": hi, "names": set(cur)})
    return buckets


def sync_bn_messages_per_step(state_keys) -> Dict[str, int]:
    """SyncBatchNorm all-reduces per train step of a ResNet pose net, from its state_dict keys: one message per BatchNorm layer and
    direction, except that conv1 and the projection shortcut of a stage's first bottleneck share one forward message (same input: both
    convs are launched first) and bn3 and that shortcut share one backward message (the shortcut's dy is bn3's g).  52 + 52 for
    ResNet50-DConv (56 BatchNorm layers, 4 shortcuts), 51 + 51 for the DUC head - the numbers `PoseTrainer.collective_count` reaches."""
    bn = [k[:-len(".running_mean")] for k in state_keys if k.endswith(".running_mean")]
    shortcuts = [k for k in bn if ".downsample." in k]
    return {"batchnorm_layers": len(bn), "forward": len(bn) - len(shortcuts), "backward": len(bn) - len(shortcuts),
            "per_step": 2 * (len(bn) - len(shortcuts))}


class FlatParams:
    """All parameters of `model` as views into one flat fp32 buffer (+ a flat gradient buffer of the same layout)."""

    def __init__(self, model: torch.nn.Module, attach_grads: bool = True):
        params = [(n, p) for n, p in model.named_parameters()]
        dev = params[0][1].device
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n, p in params:
            self.offsets[n] = (off, p.numel())
            off += _round_up(p.numel(), 4)          # keep every view 16-byte aligned
        self.numel = _round_up(off, 4)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self._views: Dict[tuple, tuple] = {}
        for n, p in params:
            o, k = self.offsets[n]
            self.data[o:o + k].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + k].view(p.shape)
            if attach_grads:                            # (the autograd surface leaves .grad to autograd: None until the first backward)
                p.grad = self.grad[o:o + k].view(p.shape)

    def view(self, name: str, grad: bool = False) -> torch.Tensor:
        base = self.grad if grad else self.data
        key = (name, grad)
        hit = self._views.get(key)
        if hit is not None and hit[0] is base:           # (the gradient buffer can be swapped: PoseTrainer.autograd_backward)
            return hit[1]
        o, k = self.offsets[name]
        v = base[o:o + k]
        self._views[key] = (base, v)
        return v


@dataclass
class Act:
    data: torch.Tensor                 # NHWC [B,H,W,C]
    h: int
    w: int
    c: int
    grad: Optional[torch.Tensor] = None
    needs_grad: bool = True
    consumers: int = 0                 # forward ops that read this activation (conv inputs, residual adds, shuffles)
    contrib: int = 0                   # backward: how many of them have added their share to .grad so far
    bn: Optional[tuple] = None         # produced by BatchNorm+ReLU (no residual): (z, mean, invstd) of that layer
    sibling: Optional[tuple] = None    # SyncBN: produced by a bare BatchNorm (projection shortcut): (z, mean, invstd, bn name)
    presums: Optional[tuple] = None    # SyncBN backward: global (sum g*xhat, sum g) already exchanged by the consumer's message
    bstats: Optional[tuple] = None     # backward: (partial sums [2, rows, stride], rows) left by the dgrad launch that wrote .grad
    bn2: Optional[tuple] = None        # block output with a projection shortcut: (z, mean, invstd, bn name) of the shortcut's BatchNorm, whose
                                       # dy is this output's g: its sum g * xhat rides on the same dgrad epilogue (part[2])
    lazy_ok: bool = False              # an identity Bottleneck's input: its other consumer is conv1 (1x1, stride 1), whose dgrad can take the
                                       # residual share of .grad as (dy of the block output, its ReLU bit mask) instead of a written tensor
    lazy_g: Optional[tuple] = None     # backward: that pair, left by bn3's pass for conv1's dgrad (sp_conv2d_dgrad_bn_bwd_stats_macc)
    mask: Optional[torch.Tensor] = None    # bf16 training: the ReLU bit mask of this BatchNorm+ReLU output (uint8, one byte per 8 channels)
    grad_event: Optional[object] = None    # backward: .grad's first share was written on the branch stream; whoever touches .grad next waits
    deferred: list = field(default_factory=list)   # ... and then runs these (the branch's weight-gradient jobs, queued on the main stream)
    pending_apply: Optional[tuple] = None  # forward: .data (and .mask) are NOT written yet - the one consumer, a 1x1 conv, forms relu(bn(z)) in its staging pass
                                           # and writes them (sp_conv2d_fwd_bn_stats_abn): (z, mean, invstd, gamma, beta)


@dataclass
class PackJob:
    src_name: str
    dst: torch.Tensor
    dims: tuple
    strides: tuple
    valid: tuple
    base: int
    dst_off: int = 0


class ConvT:
    """One conv / transposed-conv layer in training: forward launch, dgrad launches, wgrad, and its pack jobs."""

    def __init__(self, tr: "PoseTrainer", name: str, kind: str, weight: torch.Tensor, h: int, w: int, stride: int = 1, pad: int = 0,
                 c_in_buf: Optional[int] = None, bias_name: Optional[str] = None, out_nchw: bool = False, need_dgrad: bool = True,
                 taps_w: Optional[int] = None, groups: int = 1):
        self.tr, self.name, self.kind, self.wname = tr, name, kind, name + ".weight"
        self.groups = groups
        self.gpacks: List[tuple] = []                   # grouped layers: (transpose, taps_h, taps_w, ky0, ky_step, kx0, kx_step, dst) of sp_pack_conv_weights_grouped_taps
        self._rows_cache: Dict[tuple, int] = {}
        self.stride, self.pad, self.h, self.w = stride, pad, h, w
        self.bias_name, self.out_nchw, self.need_dgrad = bias_name, out_nchw, need_dgrad
        dev = weight.device
        self.pack_jobs: List[PackJob] = []
        self.bf16 = tr.bf16
        wdt = torch.bfloat16 if self.bf16 else torch.float32
        kmul = 64 if self.bf16 else 32                 # K tile = 128 bytes
        fbf = SP_CONV_BF16 if self.bf16 else 0
        self._wdt, self._kmul, self._fbf = wdt, kmul, fbf
        if kind == "conv" and groups > 1:
            self._init_grouped(weight, h, w, stride, pad, need_dgrad)
        elif kind == "conv":
            O, I, kh, kw = weight.shape
            self.O, self.I, self.kh, self.kw = O, I, kh, kw
            ci = c_in_buf or I                                     # stem: 3 -> NHWC4
            stem = c_in_buf is not None and c_in_buf > I
            tw = taps_w or ((_round_up(kw, 8) if kw > 4 else 4) if stem else kw)     # stem: tap rows padded so that K fills whole tiles
            self.ci, self.tw = ci, tw
            # K = (tap rows) x tw x ci must fill whole 128-byte K tiles; where it does not (bf16 with 32 channels: 9 x 32 = 288) whole zero tap
            # ROWS are appended - the pack job writes them as padding and the kernel's gather treats a tap row >= taps_h as out of range
            khp = self._pad_rows(kh, tw * ci)
            k_pad, n_pad = khp * tw * ci, n_pad_for(O)
            self.w_fwd = torch.zeros((n_pad, k_pad), dtype=wdt, device=dev)
            # fwd pack: dst [n_pad][khp][tw][ci] <- W[o][c][ty][tx]
            self.pack_jobs.append(PackJob(self.wname, self.w_fwd, (n_pad, khp, tw, ci), (I * kh * kw, kw, 1, kh * kw), (O, kh, kw, I), 0))
            self.oh, self.ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
            d = ConvDesc()
            d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, ci
            d.grid_h, d.grid_w, d.c_out, d.n_pad = self.oh, self.ow, O, n_pad
            d.taps_h, d.taps_w, d.k_pad, d.stride = kh, tw, k_pad, stride
            d.dy0, d.dy_step, d.dx0, d.dx_step = -pad, 1, -pad, 1
            d.out_h, d.out_w, d.out_c = self.oh, self.ow, O
            d.oy_mul = d.ox_mul = 1
            d.oy_add = d.ox_add = 0
            d.phases_y = d.phases_x = 1
            d.flags = (SP_CONV_OUT_NCHW if out_nchw else 0) | fbf
            self.d_fwd = d
            self.c_out_buf = O
            self.flops = 2 * self.oh * self.ow * O * I * kh * kw
            # wgrad: g = dz [M, O_buf], a = x gathered with the forward geometry
            self.d_wgrad = d
            self.wg = dict(n_valid=O, c_valid=I, kw_valid=kw, s_n=I * kh * kw, s_c=kh * kw)
            if need_dgrad:
                self._build_conv_dgrad(weight)
        elif kind == "deconv":                                   # ConvTranspose2d(k=4, s=2, p=1), weight [I,O,4,4]
            I, O, kh, kw = weight.shape
            assert (kh, kw) == (4, 4) and stride == 2 and pad == 1
            self.O, self.I, self.kh, self.kw, self.ci = O, I, 4, 4, I
            n_pad = n_pad_for(O)
            self.w_fwd = torch.zeros((4 * n_pad, 4 * I), dtype=wdt, device=dev)
            for py in range(2):
                for px in range(2):
                    ph = py * 2 + px                               # W[ci][co][2ty+1-py][2tx+1-px] -> [n][ty][tx][ci]
                    self.pack_jobs.append(PackJob(self.wname, self.w_fwd, (n_pad, 2, 2, I), (16, 8, 2, O * 16), (O, 2, 2, I),
                                                  (1 - py) * 4 + (1 - px), ph * n_pad * 4 * I))
            self.oh, self.ow = 2 * h, 2 * w
            d = ConvDesc()
            d.batch, d.in_h, d.in_w, d.c_in = 1, h, w, I
            d.grid_h, d.grid_w, d.c_out, d.n_pad = h, w, O, n_pad
            d.taps_h, d.taps_w, d.k_pad, d.stride = 2, 2, 4 * I, 1
            d.dy0, d.dy_step, d.dx0, d.dx_step = 0, -1, 0, -1
            d.out_h, d.out_w, d.out_c = 2 * h, 2 * w, O
            d.oy_mul, d.oy_add, d.ox_mul, d.ox_add = 2, 0, 2, 0
            d.phases_y = d.phases_x = 2
            d.flags = fbf
            self.d_fwd = d
            self.c_out_buf = O
            self.flops = 2 * h * w * I * O * 16
            # dgrad = Conv2d(k=4, s=2, p=1) of dy with Wd[ci][(ky,kx,co)] = W[ci][co][ky][kx]
            nd = n_pad_for(I)
            self.w_dgrad = [torch.zeros((nd, 16 * O), dtype=wdt, device=dev)]
            self.pack_jobs.append(PackJob(self.wname, self.w_dgrad[0], (nd, 4, 4, O), (O * 16, 4, 1, 16), (I, 4, 4, O), 0))
            g = ConvDesc()
            g.batch, g.in_h, g.in_w, g.c_in = 1, 2 * h, 2 * w, O
            g.grid_h, g.grid_w, g.c_out, g.n_pad = h, w, I, nd
            g.taps_h, g.taps_w, g.k_pad, g.stride = 4, 4, 16 * O, 2
            g.dy0, g.dy_step, g.dx0, g.dx_step = -1, 1, -1, 1
            g.out_h, g.out_w, g.out_c = h, w, I
            g.oy_mul = g.ox_mul = 1
            g.oy_add = g.ox_add = 0
            g.phases_y = g.phases_x = 1
            g.flags = fbf | (SP_CONV_OUT_F32 if (self.bf16 and not tr.g16) else 0)   # activation gradients: fp32 unless grad_dtype is bf16
            self.d_dgrad = [g]
            self.dgrad_full_cover = True
            # wgrad: dW[ci][co][ky][kx] = sum_m x[m][ci] * dy[(2iy-1+ky, 2ix-1+kx)][co]: g = x, a = dy gathered like the dgrad conv
            self.d_wgrad = g
            self.wg = dict(n_valid=I, c_valid=O, kw_valid=4, s_n=O * 16, s_c=16)
        else:
            raise ValueError(kind)

    def _init_grouped(self, weight, h, w, stride, pad, need_dgrad) -> None:
        """nn.Conv2d(groups = g) with c_out == c_in (the 3x3 of the resnext* Bottlenecks, nets/pose_resnet_dconv.py:97-101): forward and input
        gradient are grouped launches of the implicit GEMM (sp_conv_desc.c_in_group: an N tile of `panel` channels reads only its own groups'
        channels; block-diagonal panels from sp_pack_conv_weights_grouped_taps), the weight gradient a streaming reduction of its own
        (sp_conv2d_wgrad_grouped).  No statistics epilogues: the BatchNorm passes around it take their sums from the tensors."""
        C, cpg, kh, kw = weight.shape
        dev, wdt, fbf = weight.device, self._wdt, self._fbf
        if C % self.groups or C // self.groups != cpg or kh * kw > 9 or 256 % cpg:
            raise NotImplementedError(f"{self.name}: grouped convolutions are lowered for c_out == c_in with a group width dividing 256 and at most 9 taps "
                                      f"(weight {tuple(weight.shape)}, groups {self.groups})")
        panel = 64
        while panel % cpg:
            panel *= 2
        if C % panel or panel > 128:
            raise NotImplementedError(f"{self.name}: no panel width for {self.groups} groups of {cpg} channels in {C}")
        self.O, self.I, self.kh, self.kw, self.ci, self.tw, self.panel = C, cpg, kh, kw, C, kw, panel
        self.oh, self.ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        self.c_out_buf = C
        self.flops = 2 * self.oh * self.ow * C * cpg * kh * kw
        self.one_launch_phases = False                   # (sp_conv2d_dgrad_phases has no grouped form: one launch per output phase)

        def desc(in_h, in_w, gh, gw, th, tw, st, dy0, dys, dx0, dxs, oh, ow, oym=1, oya=0, oxm=1, oxa=0, out_f32=False):
            d = ConvDesc()
            d.batch, d.in_h, d.in_w, d.c_in = 1, in_h, in_w, C
            d.grid_h, d.grid_w, d.c_out, d.n_pad = gh, gw, C, C
            d.taps_h, d.taps_w, d.k_pad, d.stride = th, tw, th * tw * panel, st
            d.dy0, d.dy_step, d.dx0, d.dx_step = dy0, dys, dx0, dxs
            d.out_h, d.out_w, d.out_c = oh, ow, C
            d.oy_mul, d.oy_add, d.ox_mul, d.ox_add = oym, oya, oxm, oxa
            d.phases_y = d.phases_x = 1
            d.flags = fbf | (SP_CONV_OUT_F32 if out_f32 else 0)
            d.c_in_group, d.tile_m, d.tile_n = panel, 128, panel
            return d

        self.w_fwd = torch.zeros((C, kh * kw * panel), dtype=wdt, device=dev)
        self.gpacks.append((0, kh, kw, 0, 1, 0, 1, self.w_fwd))
        self.d_fwd = desc(h, w, self.oh, self.ow, kh, kw, stride, -pad, 1, -pad, 1, self.oh, self.ow)
        self.d_wgrad = self.d_fwd
        self.w_dgrad, self.d_dgrad = [], []
        self.dgrad_full_cover = True
        if not need_dgrad:
            return
        g32 = self.bf16 and not self.tr.g16                 # activation gradients: fp32 unless grad_dtype is bf16
        if stride == 1:
            wd = torch.zeros((C, kh * kw * panel), dtype=wdt, device=dev)
            self.gpacks.append((1, kh, kw, kh - 1, -1, kw - 1, -1, wd))          # Wd[c][(ty, tx, o)] = W[o][c][kh-1-ty][kw-1-tx]
            pp = kh - 1 - pad
            self.w_dgrad.append(wd)
            self.d_dgrad.append(desc(self.oh, self.ow, h, w, kh, kw, 1, -pp, 1, -pp, 1, h, w, out_f32=g32))
        else:
            assert stride == 2 and h % 2 == 0 and w % 2 == 0
            for py in range(2):                             # one launch per output phase (ConvT._build_conv_dgrad's decomposition)
                for px in range(2):
                    ky0, kx0 = (py + pad) % 2, (px + pad) % 2
                    th, tw = len(range(ky0, kh, 2)), len(range(kx0, kw, 2))
                    if th == 0 or tw == 0:
                        self.dgrad_full_cover = False
                        continue
                    wd = torch.zeros((C, th * tw * panel), dtype=wdt, device=dev)
                    self.gpacks.append((1, th, tw, ky0, 2, kx0, 2, wd))
                    self.w_dgrad.append(wd)
                    self.d_dgrad.append(desc(self.oh, self.ow, h // 2, w // 2, th, tw, 1, (py + pad - ky0) // 2, -1, (px + pad - kx0) // 2, -1, h, w,
                                             2, py, 2, px, out_f32=g32))

    def pack_grouped(self, stream) -> None:
        """Regenerate this grouped layer's block-diagonal panels (forward + input-gradient phases) from the flat parameter buffer."""
        lib, w = _lib.lib(), self.tr.flat.view(self.wname)
        for tr_, th, tw, ky0, kys, kx0, kxs, dst in self.gpacks:
            _lib.check(lib.sp_pack_conv_weights_grouped_taps(P(w), self.O, self.groups, self.kh, self.kw, tr_, th, tw, ky0, kys, kx0, kxs, self.panel, P(dst),
                                                             int(self.bf16), stream), self.name + ".pack")

    def wgrad_grouped(self, x: torch.Tensor, dz: torch.Tensor, B: int, stream=None) -> None:
        """dW of a grouped layer, written into the flat gradient buffer (sp_conv2d_wgrad_grouped: partial sums per row chunk, fixed-order fold)."""
        lib, tr = _lib.lib(), self.tr
        need = ctypes.c_int64(0)
        _lib.check(lib.sp_conv2d_wgrad_grouped_workspace(B, self.oh, self.O, self.groups, self.kh, self.kw, ctypes.byref(need)), self.name + ".wgrad")
        ws = self._new(((need.value + 3) // 4,), torch.float32, x.device)
        done = self._timed("wgrad")
        _lib.check(lib.sp_conv2d_wgrad_grouped(P(x), P(dz), int(self.bf16), B, self.h, self.w, self.oh, self.ow, self.O, self.groups, self.kh, self.kw,
                                               self.stride, self.pad, P(tr.flat.view(self.wname, grad=True)), P(ws), ws.numel() * 4,
                                               stream if stream is not None else _lib.current_stream()), self.name + ".wgrad")
        done()

    def _pad_rows(self, rows: int, row_elems: int) -> int:
        """Smallest number of tap rows >= `rows` whose elements fill whole K tiles."""
        r = rows
        while (r * row_elems) % self._kmul:
            r += 1
        return r

    # dgrad of a Conv2d: stride 1 -> one conv with flipped taps; stride 2 -> one launch per output phase
    def _build_conv_dgrad(self, weight):
        O, I, kh, kw, s, p = self.O, self.I, self.kh, self.kw, self.stride, self.pad
        dev = weight.device
        wdt, kmul, fbf = self._wdt, self._kmul, self._fbf
        nd = n_pad_for(I)
        # channels of the incoming gradient buffer: 17 heat-map channels -> one whole K tile (32 fp32 / 64 bf16); a channel count that is a
        # whole number of 16-byte chunks stays as it is (32 channels in bf16) and zero tap rows fill the K tile instead (_pad_rows)
        epc = 8 if self.bf16 else 4
        Ob = O if (O % kmul == 0 or O % epc == 0) else _round_up(O, kmul)
        self.c_out_buf = Ob
        self.w_dgrad, self.d_dgrad = [], []
        if s == 1:
            khp = self._pad_rows(kh, kw * Ob)
            wd = torch.zeros((nd, khp * kw * Ob), dtype=wdt, device=dev)
            # Wd[c][(ty,tx,o)] = W[o][c][kh-1-ty][kw-1-tx]
            self.pack_jobs.append(PackJob(self.wname, wd, (nd, khp, kw, Ob), (kh * kw, -kw, -1, I * kh * kw), (I, kh, kw, O),
                                          (kh - 1) * kw + (kw - 1)))
            g = ConvDesc()
            g.batch, g.in_h, g.in_w, g.c_in = 1, self.oh, self.ow, Ob
            g.grid_h, g.grid_w, g.c_out, g.n_pad = self.h, self.w, I, nd
            g.taps_h, g.taps_w, g.k_pad, g.stride = kh, kw, khp * kw * Ob, 1
            pp = kh - 1 - p
            g.dy0, g.dy_step, g.dx0, g.dx_step = -pp, 1, -pp, 1
            g.out_h, g.out_w, g.out_c = self.h, self.w, I
            g.oy_mul = g.ox_mul = 1
            g.oy_add = g.ox_add = 0
            g.phases_y = g.phases_x = 1
            g.flags = fbf | (SP_CONV_OUT_F32 if (self.bf16 and not self.tr.g16) else 0)
            self.w_dgrad.append(wd); self.d_dgrad.append(g)
            self.dgrad_full_cover = True
        else:
            assert s == 2 and self.h % 2 == 0 and self.w % 2 == 0
            # dx[2g+py] = sum over ky with (2g+py+p-ky) even: oy = (2g+py+p-ky)/2.  Tap t of phase py: ky = ky0 + 2t (ky < kh),
            # oy = g + (py+p-ky0)/2 - t  ->  dy0 = (py+p-ky0)/2, dy_step = -1
            self.dgrad_full_cover = True
            for py in range(2):
                for px in range(2):
                    ky0, kx0 = (py + p) % 2, (px + p) % 2
                    th, tw = len(range(ky0, kh, 2)), len(range(kx0, kw, 2))
                    if th == 0 or tw == 0:
                        self.dgrad_full_cover = False             # 1x1 stride 2: only phase (0,0) receives gradient
                        continue
                    thp = self._pad_rows(th, tw * Ob)
                    wd = torch.zeros((nd, thp * tw * Ob), dtype=wdt, device=dev)
                    self.pack_jobs.append(PackJob(self.wname, wd, (nd, thp, tw, Ob), (kh * kw, 2 * kw, 2, I * kh * kw), (I, th, tw, O),
                                                  ky0 * kw + kx0))
                    g = ConvDesc()
                    g.batch, g.in_h, g.in_w, g.c_in = 1, self.oh, self.ow, Ob
                    g.grid_h, g.grid_w, g.c_out, g.n_pad = self.h // 2, self.w // 2, I, nd
                    g.taps_h, g.taps_w, g.k_pad, g.stride = th, tw, thp * tw * Ob, 1
                    g.dy0, g.dy_step, g.dx0, g.dx_step = (py + p - ky0) // 2, -1, (px + p - kx0) // 2, -1
                    g.out_h, g.out_w, g.out_c = self.h, self.w, I
                    g.oy_mul, g.oy_add, g.ox_mul, g.ox_add = 2, py, 2, px
                    g.phases_y = g.phases_x = 1
                    g.flags = fbf | (SP_CONV_OUT_F32 if (self.bf16 and not self.tr.g16) else 0)
                    self.w_dgrad.append(wd); self.d_dgrad.append(g)

    # ---- launches ----
    def _stats_rows(self, d, B: int, what: str) -> int:
        """Partial rows the STATS / BSTATS epilogue of a launch of `d` writes (a pure function of the grid and the tile: asked once)."""
        key = (id(d), B, d.tile_m, d.tile_n)
        r = self._rows_cache.get(key)
        if r is None:
            d.batch = B
            c = ctypes.c_int(0)
            _lib.check(_lib.lib().sp_conv2d_bn_stats_rows(d, ctypes.byref(c)), what)
            r = self._rows_cache[key] = c.value
        return r

    def _new(self, shape, dtype, device, zero: bool = False) -> torch.Tensor:
        """A result / scratch tensor: from the trainer's step arena when it has one (PoseTrainer._take), else a fresh allocation."""
        take = getattr(self.tr, "_take", None)
        if take is not None:
            return take(tuple(shape), dtype, device, zero)
        return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)

    def _timed(self, kind: str, stream=None):
        """bench.py's per-kernel roofline: when the trainer collects kernel events, bracket this launch family with HIP events on the
        stream it is issued to.  Returns the closing callback (a no-op otherwise)."""
        ke = self.tr.kernel_events
        if ke is None:
            return lambda: None
        st = stream if stream is not None else torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        ke.append((kind, self.name, self.flops, e0, e1))
        return lambda: e1.record(st)

    def forward(self, x: torch.Tensor, B: int, out: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                relu: bool = False) -> torch.Tensor:
        lib, d = _lib.lib(), self.d_fwd
        d.batch = B
        if out is None:
            out = self._new((B, d.out_h, d.out_w, d.out_c), self._wdt, x.device)
        done = self._timed("forward")
        keep = d.flags
        if relu:
            d.flags = keep | SP_CONV_RELU
        try:
            _lib.check(lib.sp_conv2d_fwd(d, P(x), P(self.w_fwd), None, P(shift), None, P(out), _lib.current_stream()), self.name)
        finally:
            d.flags = keep
        done()
        return out

    def forward_bn_stats(self, x: torch.Tensor, B: int):
        """Forward launch whose epilogue also leaves the per-channel partial sums BatchNorm needs (no extra pass over z).
        Returns (z, partial sums [rows, n_pad], partial sums of squares, rows)."""
        lib, d = _lib.lib(), self.d_fwd
        d.batch = B
        rows = self._stats_rows(d, B, self.name)
        part = self._new((2, rows, d.n_pad), torch.float32, x.device)
        out = self._new((B, d.out_h, d.out_w, d.out_c), self._wdt, x.device)
        done = self._timed("forward")
        _lib.check(lib.sp_conv2d_fwd_bn_stats(d, P(x), P(self.w_fwd), P(out), P(part[0]), P(part[1]), rows, _lib.current_stream()),
                   self.name)
        done()
        return out, part, rows

    def abn_ok(self, B: int) -> bool:
        """Can this layer's forward take relu(BatchNorm(z)) of the previous layer in its staging pass (sp_conv2d_fwd_bn_stats_abn)?  A bf16 1x1
        stride-1 convolution with at most 512 input channels (< 12 K tiles) on the implicit-GEMM kernel."""
        d = self.d_fwd
        return (self.bf16 and self.kind == "conv" and (self.kh, self.kw) == (1, 1) and self.stride == 1 and self.pad == 0 and d.c_in <= 512
                and d.c_in % 64 == 0 and d.k_pad // 64 < 12 and d.kernel == _lib.SP_CONV_KERNEL_IGEMM and not self.out_nchw)

    def forward_bn_stats_abn(self, z_in: torch.Tensor, B: int, mean, invstd, gamma, beta, y_in: torch.Tensor, mask_in: Optional[torch.Tensor]):
        """forward_bn_stats with x = relu(BatchNorm(z_in)) formed in the kernel's staging pass; also writes x (`y_in`) and its ReLU bit mask."""
        lib, d = _lib.lib(), self.d_fwd
        d.batch = B
        rows = self._stats_rows(d, B, self.name)
        part = self._new((2, rows, d.n_pad), torch.float32, z_in.device)
        out = self._new((B, d.out_h, d.out_w, d.out_c), self._wdt, z_in.device)
        done = self._timed("forward")
        _lib.check(lib.sp_conv2d_fwd_bn_stats_abn(d, P(z_in), P(mean), P(invstd), P(gamma), P(beta), P(y_in), P(mask_in), P(self.w_fwd), P(out),
                                                  P(part[0]), P(part[1]), rows, _lib.current_stream()), self.name)
        done()
        return out, part, rows

    def dgrad(self, dz: torch.Tensor, B: int, acc: Optional[torch.Tensor], bn_src: Optional["Act"] = None,
              acc_masked: Optional[tuple] = None) -> torch.Tensor:
        """dx (+= into `acc` when given).  `bn_src`: the activation dx is the gradient of, when it came out of a BatchNorm(+residual)+ReLU
        and this launch family is the LAST of its consumers to contribute: the epilogue then holds the complete dy and also reduces
        that layer's backward sums (bn_src.bstats)."""
        lib = _lib.lib()
        if bn_src is not None:
            d0 = self.d_dgrad[0]
            if acc is not None:
                # the other consumers' shares are already in `acc`: this launch family adds the last one in place, so its epilogue sees
                # the COMPLETE dy (every element must be written by it: stride-2 1x1 families leave holes and are not offered here)
                assert self.dgrad_full_cover
                dx = acc
            else:
                dx = self._new((B, d0.out_h, d0.out_w, d0.out_c), self.tr.grad_dtype, dz.device, zero=not self.dgrad_full_cover)
            one = len(self.d_dgrad) > 1 and self.dgrad_full_cover and self.one_launch_phases
            if one and d0.tile_m == 0:
                d0.batch = B
                tm, tn = ctypes.c_int(0), ctypes.c_int(0)
                _lib.check(lib.sp_conv2d_default_tile(d0, ctypes.byref(tm), ctypes.byref(tn)), self.name + ".dgrad")
                d0.tile_m, d0.tile_n = tm.value, tn.value
            need = []
            for d in self.d_dgrad:
                d.batch = B
                if one:
                    d.tile_m, d.tile_n = d0.tile_m, d0.tile_n      # one launch: one tile shape (and its partial-row count) for every phase
                need.append(self._stats_rows(d, B, self.name + ".dgrad"))
            total, stride = sum(need), self.d_dgrad[0].n_pad
            two = bn_src.bn2 is not None
            part = self._new((3 if two else 2, total, stride), torch.float32, dz.device)
            z, mean, invstd = bn_src.bn
            row0 = 0
            ysrc, mflag = bn_src.data, 0
            if bn_src.mask is not None and self.tr.g16:
                ysrc, mflag = bn_src.mask, SP_CONV_BN_Y_MASK        # the ReLU bit mask instead of y: 1/16 of the bytes
            keep_flags = [d.flags for d in self.d_dgrad]
            for d in self.d_dgrad:
                d.flags |= mflag
            try:
                if acc_masked is not None:
                    assert len(self.d_dgrad) == 1 and acc is None and self.tr.g16
                    d = self.d_dgrad[0]
                    z2, mean2, invstd2 = (bn_src.bn2[:3] if two else (None, None, None))
                    done = self._timed("dgrad")
                    _lib.check(lib.sp_conv2d_dgrad_bn_bwd_stats_macc(d, P(dz), P(self.w_dgrad[0]), P(acc_masked[0]), P(acc_masked[1]), P(dx), P(ysrc),
                                                                     P(z), P(mean), P(invstd), P(part[0]), P(part[1]), P(z2), P(mean2), P(invstd2),
                                                                     P(part[2]) if two else None, total, _lib.current_stream()), self.name + ".dgrad")
                    done()
                    bn_src.bstats = (part, total)
                    return dx
                return self._dgrad_bstats(lib, dz, B, acc, dx, bn_src, ysrc, z, mean, invstd, part, total, need, one, two)
            finally:
                for d, f in zip(self.d_dgrad, keep_flags):
                    d.flags = f
        return self._dgrad_plain(lib, dz, B, acc)

    def _dgrad_bstats(self, lib, dz, B, acc, dx, bn_src, ysrc, z, mean, invstd, part, total, need, one, two):
        if True:
            row0 = 0
            done = self._timed("dgrad")
            if one:
                # a stride-2 conv's output phases (different tap counts) as ONE launch: same rows, same bits, three launch boundaries less
                descs, ws_ = self._phase_arrays()
                z2, mean2, invstd2 = (bn_src.bn2[:3] if two else (None, None, None))
                _lib.check(lib.sp_conv2d_dgrad_phases(descs, len(self.d_dgrad), P(dz), ws_, P(acc), P(dx), P(ysrc), P(z), P(mean), P(invstd),
                                                      P(part[0]), P(part[1]), P(z2), P(mean2), P(invstd2), P(part[2]) if two else None, total,
                                                      _lib.current_stream()), self.name + ".dgrad")
                done()
                bn_src.bstats = (part, total)
                return dx
            for d, w, n in zip(self.d_dgrad, self.w_dgrad, need):
                if two:
                    z2, mean2, invstd2, _ = bn_src.bn2
                    _lib.check(lib.sp_conv2d_dgrad_bn_bwd_stats2(d, P(dz), P(w), P(acc), P(dx), P(ysrc), P(z), P(mean), P(invstd),
                                                                 P(part[0, row0:]), P(part[1, row0:]), P(z2), P(mean2), P(invstd2),
                                                                 P(part[2, row0:]), n, _lib.current_stream()), self.name + ".dgrad")
                else:
                    _lib.check(lib.sp_conv2d_dgrad_bn_bwd_stats(d, P(dz), P(w), P(acc), P(dx), P(ysrc), P(z), P(mean), P(invstd),
                                                                P(part[0, row0:]), P(part[1, row0:]), n, _lib.current_stream()),
                               self.name + ".dgrad")
                row0 += n
            done()
            bn_src.bstats = (part, total)
            return dx

    def _dgrad_plain(self, lib, dz, B, acc):
        if acc is None:
            d0 = self.d_dgrad[0]
            shape = (B, d0.out_h, d0.out_w, d0.out_c)
            acc_t = self._new(shape, self.tr.grad_dtype, dz.device, zero=not self.dgrad_full_cover)
            res = None
        else:
            acc_t, res = acc, acc
        done = self._timed("dgrad")
        if len(self.d_dgrad) > 1 and self.dgrad_full_cover and self.one_launch_phases:
            for d in self.d_dgrad:
                d.batch = B
            descs, ws_ = self._phase_arrays()
            _lib.check(lib.sp_conv2d_dgrad_phases(descs, len(self.d_dgrad), P(dz), ws_, P(res), P(acc_t), None, None, None, None, None, None, None,
                                                  None, None, None, 0, _lib.current_stream()), self.name + ".dgrad")
            done()
            return acc_t
        for d, w in zip(self.d_dgrad, self.w_dgrad):
            d.batch = B
            _lib.check(lib.sp_conv2d_fwd(d, P(dz), P(w), None, None, P(res), P(acc_t), _lib.current_stream()), self.name + ".dgrad")
        done()
        return acc_t

    one_launch_phases = os.environ.get("SP_PHASES_ONE_LAUNCH", "1") != "0"        # (env: development knob)

    def _phase_arrays(self):
        """The phase descriptors as one contiguous C array (+ their packed-weight pointers) for sp_conv2d_dgrad_phases; all phases run the
        tile of phase 0 (the 2x2-tap phase, three quarters of the work)."""
        n = len(self.d_dgrad)
        arr = (ConvDesc * n)()
        for i, d in enumerate(self.d_dgrad):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(d), ctypes.sizeof(ConvDesc))
            arr[i].tile_m, arr[i].tile_n = self.d_dgrad[0].tile_m, self.d_dgrad[0].tile_n
        ws_ = (_lib.c_void_p * n)(*[w.data_ptr() for w in self.w_dgrad])
        return arr, ws_

    def wgrad_job(self, x: torch.Tensor, dz: torch.Tensor, B: int, job: Optional["_lib.WgradJob"] = None) -> "_lib.WgradJob":
        """This layer's record of a sp_conv2d_wgrad_batched call (conv: g = dz, a = x; transposed conv: g = x, a = dz)."""
        job = job if job is not None else _lib.WgradJob()
        self.d_wgrad.batch = B
        ctypes.memmove(ctypes.byref(job.desc), ctypes.byref(self.d_wgrad), ctypes.sizeof(ConvDesc))
        g, a = (dz, x) if self.kind == "conv" else (x, dz)
        job.g, job.a, job.dw = g.data_ptr(), a.data_ptr(), self.tr.flat.view(self.wname, grad=True).data_ptr()
        job.g_channels, job.n_valid, job.c_valid, job.kw_valid = g.shape[-1], self.wg["n_valid"], self.wg["c_valid"], self.wg["kw_valid"]
        job.dst_stride_n, job.dst_stride_c = self.wg["s_n"], self.wg["s_c"]
        return job

    def wgrad(self, x: torch.Tensor, dz: torch.Tensor, B: int, stream=None):
        lib, tr = _lib.lib(), self.tr
        d = self.d_wgrad
        d.batch = B
        g, a = (dz, x) if self.kind == "conv" else (x, dz)
        gc = g.shape[-1]
        done = self._timed("wgrad", tr._wgrad_stream if stream is not None else None)
        _lib.check(lib.sp_conv2d_wgrad(d, P(g), gc, P(a), self.wg["n_valid"], self.wg["c_valid"], self.wg["kw_valid"], self.wg["s_n"],
                                       self.wg["s_c"], P(tr.flat.view(self.wname, grad=True)), P(tr.wgrad_ws), tr.wgrad_ws.numel() * 4,
                                       stream if stream is not None else _lib.current_stream()), self.name + ".wgrad")
        done()


class PoseTrainer:
    """Train step (fp32 or bf16 compute) for `simple_pose_amd.nets.pose_resnet_dconv.ResNet` / `pose_resnet_duc.ResNet` - the two
    models the DDP solver builds (ddp...:65-68) - on one GPU per process (+ optional process group)."""

    def __init__(self, model, in_h: int = 256, in_w: int = 192, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, dtype: str = "fp32", sync_bn: Optional[bool] = None, bucket_mb: float = 32.0,
                 broadcast_init: bool = True, overlap_wgrad: bool = True, collectives: bool = True, sync_bn_latency_us: float = 0.0,
                 native_comm: Optional[bool] = None, sync_bn_inline: bool = True, grad_dtype: Optional[str] = None):
        """dtype "bf16": bf16 activations and packed weights (v_mfma_f32_32x32x16_bf16 for forward, dgrad and wgrad), the
        BN-input gradient dz that feeds the MFMAs is bf16; fp32 master weights / weight gradients / BN statistics / Adam.
        The reference's `optim.amp` mode (ddp...:121-127) without a GradScaler (bf16 keeps fp32's exponent range).
        `grad_dtype` - the activation gradients (dy of every conv / block output): "bf16" = what autocast keeps (the default with
        dtype "bf16" on the plain DConv / DUC nets: the dgrad launches store bf16, the residual share accumulates in bf16, the BatchNorm
        backward reads it rounded; 2.2 GB less traffic per 32-image step), "fp32" = rounds 1-3 (dy stays fp32 until the BatchNorm
        backward has subtracted its per-channel means; the only setting for the SELayer nets and HRNet, whose backward kernels read fp32).
        Both sit inside the same bar against the reference-style AMP oracle (tests/test_gpu_train.py).

        With a process group of W > 1 ranks (one process per GPU):
          * parameters and BN buffers are broadcast from rank 0 at construction (DistributedDataParallel's init, ddp...:91-93);
          * `sync_bn` (default: on, as `SyncBatchNorm.convert_sync_batchnorm` at ddp...:89-90): batch statistics and the two
            backward sums of every BN layer are summed over ranks (one [2C] all-reduce each way per layer);
          * the flat gradient buffer is all-reduced in `bucket_mb`-sized contiguous slices, each launched (async, on RCCL's
            own stream) as soon as backward has produced its last gradient - final_layer's end of the buffer first.
        `native_comm` (opt-in: True, or SP_NATIVE_COMM=1; needs an nccl group and librccl): the step's collectives go to RCCL directly
        (`sp_comm_allreduce_sum_f32` / `_f64`, csrc/comm.hip) - a SyncBatchNorm message is then ONE host call that enqueues the all-reduce on the
        compute stream (the chain of dependent launches leaves nothing to run under it; through torch.distributed each message costs
        five stream / event calls from Python and the step becomes host-bound), a gradient bucket one call on the optimizer stream.
        `sync_bn_inline`: where the emulated message latency (`sync_bn_latency_us`) is spent - on the compute stream (as the native
        path does) or on a message stream fenced by events (as torch.distributed does)."""
        if dtype not in ("fp32", "bf16"):
            raise ValueError(dtype)
        self.bf16 = dtype == "bf16"
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float32
        if getattr(model, "HEAD", None) not in ("dconv", "duc", "hrnet"):
            raise NotImplementedError("PoseTrainer lowers the ResNet DConv / DUC nets (the DDP solver's models, ddp...:65-68) and HRNet")
        self.head = model.HEAD
        # dtype of the activation gradients (dy of every conv output / block output).  "fp32": the round 1-3 behaviour.  "bf16" (with
        # dtype "bf16" only; the default there for the nets whose backward kernels all read it): what torch's autocast keeps - the
        # dgrad launches store bf16 and accumulate the residual share in bf16, the BatchNorm backward reads it rounded.
        has_se = any(".se." in k for k in model.state_dict())
        can16 = self.bf16 and self.head in ("dconv", "duc") and not has_se
        import os
        if grad_dtype is None:
            grad_dtype = os.environ.get("SP_GRAD_DTYPE") or ("bf16" if can16 else "fp32")
            if grad_dtype == "bf16" and not can16:
                grad_dtype = "fp32"
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError(grad_dtype)
        if grad_dtype == "bf16" and not can16:
            raise NotImplementedError("grad_dtype='bf16' needs dtype='bf16' and a plain ResNet DConv / DUC net (the SELayer / HRNet backward kernels "
                                      "read fp32 gradients)")
        self.g16 = grad_dtype == "bf16"
        self.grad_dtype = torch.bfloat16 if self.g16 else torch.float32
        if getattr(model, "BLOCK", "bottleneck") not in ("bottleneck", "basic"):
            raise NotImplementedError(f"PoseTrainer lowers the Bottleneck / BasicBlock ResNets and HRNet, not block type {model.BLOCK!r}")
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.pg = process_group
        self.overlap_wgrad = overlap_wgrad
        if os.environ.get("SP_FOLD_ROWS"):                              # (development knob)
            self.fold_in_consumer_rows = int(os.environ["SP_FOLD_ROWS"])
        self._wgrad_stream = None
        self._branch_stream = None
        self.overlap_shortcut = os.environ.get("SP_BRANCH", "1") != "0"        # (env: development knob)
        self._opt_stream = None
        self._opt_in_backward = False
        import torch.distributed as dist
        # collectives=False: a trainer that only computes local gradients (the autograd surface `model(x)` of nets.*: there the
        # caller - e.g. torch's DistributedDataParallel wrapper, as in ddp...:91-93 - owns the gradient exchange)
        self.world = dist.get_world_size(self.pg) if (collectives and dist.is_available() and dist.is_initialized()) else 1
        self.sync_bn = (self.world > 1) if sync_bn is None else (bool(sync_bn) and self.world > 1)
        # measurement aid (bench.py --sync-bn-latency-us): every SyncBatchNorm message additionally costs this many microseconds on a
        # side stream the consumer waits for; on ONE rank it switches the SyncBatchNorm code path on with nothing to exchange, which
        # bounds what the 104 messages per step would expose on xGMI without an 8-GPU node
        self.sync_bn_latency_us = float(sync_bn_latency_us)
        self.sync_bn_inline = bool(sync_bn_inline)
        if self.sync_bn_latency_us > 0 and (sync_bn is None or sync_bn):
            self.sync_bn = True
        self._comm = self._comm_grad = None
        self._native_comm_wanted = native_comm
        self.flat = FlatParams(model, attach_grads=collectives)
        dev = self.flat.data.device
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.step_count = 0
        self.red_ws = torch.empty(4 << 20, dtype=torch.uint8, device=dev)                  # SP_REDUCE_WORKSPACE_BYTES
        self.wgrad_ws = torch.empty(48 * 1024 * 1024, dtype=torch.float32, device=dev)    # 192 MB of split slabs
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.mse_ws = torch.empty(4096, dtype=torch.uint8, device=dev)
        self.sd = dict(model.named_parameters())
        self.buffers = dict(model.named_buffers())
        self.in_h, self.in_w = in_h, in_w
        self.layers: Dict[str, ConvT] = {}
        self._build(in_h, in_w)
        self._grouped_layers = [L for L in self.layers.values() if L.groups > 1]
        if self.world > 1 and broadcast_init:
            dist.broadcast(self.flat.data, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            for b in self.buffers.values():
                dist.broadcast(b, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        self._plan_buckets(bucket_mb)
        self.repack()
        if self.world > 1:
            self._open_native_comm()

    def _open_native_comm(self) -> None:
        """One RCCL communicator of our own over the ranks of the process group (the 128-byte id travels through the group itself)."""
        import torch.distributed as dist
        want = self._native_comm_wanted
        lib = _lib.lib()
        usable = dist.get_backend(self.pg) == "nccl" and bool(lib.sp_comm_available())
        if want is None:
            # opt-in (native_comm=True, or SP_NATIVE_COMM=1 in the environment) until the two private communicators have met a peer on a
            # multi-GPU box: `tests/test_gpu_train.py::test_two_rank_rccl_*` is that test and skips on one GPU
            import os
            want = usable and os.environ.get("SP_NATIVE_COMM", "0") == "1"
        if not want:
            return
        if not usable:
            raise _lib.HipLibraryError("native_comm=True needs an nccl-backed process group and librccl (sp_comm_available() == 1)")
        dev = self.flat.data.device
        world, rank = dist.get_world_size(self.pg), dist.get_rank(self.pg)

        def make():
            idt = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (ctypes.c_ubyte * 128)()
                _lib.check(lib.sp_comm_unique_id(buf), "sp_comm_unique_id")
                idt = torch.tensor(list(buf), dtype=torch.uint8)
            idt = idt.to(dev)
            if world > 1:
                dist.broadcast(idt, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            raw = bytes(idt.cpu().tolist())
            comm = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(lib.sp_comm_create(raw, world, rank, ctypes.byref(comm)), "sp_comm_create")
            return comm
        # TWO communicators: RCCL orders the operations of one communicator one after the other whatever stream they are issued to, so a
        # SyncBatchNorm message (compute stream, on the critical chain) would queue behind a 32 MB gradient bucket (optimizer stream) of the
        # same communicator - which is what happens to every collective of a torch.distributed process group (one communicator, one stream)
        self._comm = make()
        self._comm_grad = make()

    def close(self) -> None:
        """Release the RCCL communicator (if any); the trainer must not step afterwards."""
        if self._comm is not None:
            torch.cuda.synchronize(self.flat.data.device)
            for c in (self._comm, self._comm_grad):
                _lib.check(_lib.lib().sp_comm_destroy(c), "sp_comm_destroy")
            self._comm = self._comm_grad = None

    def _all_reduce_sum(self, t: torch.Tensor, stream, grad: bool = False) -> None:
        """In-place SUM over the ranks on `stream` through our own communicators (`grad`: the gradient buckets' one).  The element type
        of the RCCL call follows the tensor: fp32 (gradient buckets, backward SyncBatchNorm sums) or fp64 (the forward SyncBatchNorm
        message: per-channel (sum, sum of squares) in double); anything else is refused."""
        lib, comm = _lib.lib(), (self._comm_grad if grad else self._comm)
        if not t.is_contiguous():
            raise ValueError("native all-reduce needs a contiguous buffer")
        if t.dtype == torch.float32:
            _lib.check(lib.sp_comm_allreduce_sum_f32(comm, P(t), t.numel(), stream), "all-reduce f32")
        elif t.dtype == torch.float64:
            _lib.check(lib.sp_comm_allreduce_sum_f64(comm, P(t), t.numel(), stream), "all-reduce f64")
        else:
            raise TypeError(f"native all-reduce: unsupported dtype {t.dtype} (fp32 and fp64 buffers only)")

    # ---- gradient buckets (DDP reducer, reverse parameter order) ---------------------------------------------------------
    def _plan_buckets(self, bucket_mb: float):
        """Contiguous slices of the flat gradient buffer, cut at parameter boundaries walking from the END of the buffer (backward
        produces final_layer first).  Each bucket knows which parameter gradients it waits for."""
        self.buckets: List[dict] = plan_gradient_buckets(self.flat.offsets, self.flat.numel, bucket_mb, self.stem_bucket_params)
        self._bucket_of = {n: i for i, b in enumerate(self.buckets) for n in b["names"]}

    def _grads_ready(self, *names: str):
        """Called by the backward tape when the gradients of `names` have been enqueued AND the layer's dgrad launches (the last
        readers of its packed weights) are on the compute stream.  A bucket whose last gradient arrived goes out:
          * plain mode: its all-reduce is launched (async) - `all_reduce_grads()` waits, `optimizer_step()` follows;
          * `step()` mode (optimizer in backward): on a third stream the bucket is all-reduced, Adam updates its slice of the flat
            buffers and its packed weight copies are regenerated, all while backward continues with earlier layers."""
        fused = self._opt_in_backward
        if self.world == 1 and not fused:
            return
        import torch.distributed as dist
        for n in names:
            i = self._bucket_of[n]
            self._pending[i].discard(n)
            if self._pending[i] or self._works[i] is not None:
                continue
            self._wgrad_flush()                            # the bucket's last weight gradients may still be queued
            b = self.buckets[i]
            main = torch.cuda.current_stream()
            if not fused:
                if self._wgrad_tail is not None:           # the bucket's weight gradients come from the wgrad stream
                    main.wait_event(self._wgrad_tail)
                self._works[i] = dist.all_reduce(self.flat.grad[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                continue
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream(device=self.flat.data.device)
                self._opt_events = {}
            opt = self._opt_stream
            ev = self._opt_events.get(i)
            if ev is None:
                ev = self._opt_events[i] = (torch.cuda.Event(), torch.cuda.Event())
            ev[0].record(main)                              # dgrads of the bucket's layers are behind this point
            opt.wait_event(ev[0])
            if self._wgrad_tail is not None:
                opt.wait_event(self._wgrad_tail)
            with torch.cuda.stream(opt):
                if self._comm is not None:
                    self._all_reduce_sum(self.flat.grad[b["lo"]:b["hi"]], _lib.c_void_p(opt.cuda_stream), grad=True)
                elif self.world > 1 or self.force_collectives:
                    dist.all_reduce(self.flat.grad[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.pg)
                self._adam(slice(b["lo"], b["hi"]), 1.0 / self.world, _lib.c_void_p(opt.cuda_stream))
                self.repack(self._pack_rows_of_bucket[i], _lib.c_void_p(opt.cuda_stream), bucket=i)
            ev[1].record(opt)
            self._works[i] = ev[1]

    # ---- step arena ---------------------------------------------------------------------------------------------------------------
    # PoseTrainer.step / forward_backward request the same tensors in the same order every step (~300 activations, gradients and
    # partial-sum buffers): the arena hands out last step's tensor for the i-th request instead of going through the caching allocator
    # (3-4 us of host time each; the host needs ~5 ms to enqueue a 6.7 ms step).  Nothing of one step is read after the next one
    # starts: the step ends with the main stream joined to the weight-gradient and optimizer streams.  The autograd surface
    # (forward_tape / backward called by torch) allocates normally: there the caller decides how long a forward's tensors live.
    _arena_on = False

    def _take(self, shape, dtype, device, zero: bool = False) -> torch.Tensor:
        if not self._arena_on:
            # (inside a branch-stream section the block comes from the BRANCH stream's pool of the caching allocator: right for the section's
            # temporaries; what it hands to the main stream lives until the tape is dropped, after the join; the main-pool tensor it reads
            # and drops is record_stream'ed there)
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        a, i = self._arena, self._arena_i
        self._arena_i = i + 1
        if i < len(a):
            t, key = a[i]
            if key == (shape, dtype, device):
                if zero:
                    t.zero_()
                return t
            del a[i:]                      # the request sequence changed (another batch size): rebuild from here on
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        a.append((t, (tuple(shape), dtype, device)))
        return t

    # ---- SyncBatchNorm messages: issued where the sums exist, waited for where the statistics are consumed -----------------------
    def _exchange(self, t: torch.Tensor):
        """SUM the fp32 tensor `t` over the ranks; returns the token `_exchange_wait` takes.  Native path: the all-reduce is enqueued on
        the compute stream right here (one call; the consumer is simply the next launch).  torch.distributed path: asynchronously on
        RCCL's own stream behind the compute stream's current position - whatever is launched between the two calls runs under it."""
        self.collective_count += 1
        work = ev = None
        if self._comm is not None:
            self._all_reduce_sum(t, _lib.current_stream(t.device))
        elif self.world > 1 or self.force_collectives:
            import torch.distributed as dist
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if self.sync_bn_latency_us > 0:
            dev = t.device
            if self.sync_bn_inline:
                _lib.check(_lib.lib().sp_stream_delay_us(self.sync_bn_latency_us, _lib.current_stream(dev)), "delay")
                return work, ev
            if getattr(self, "_comm_stream", None) is None:
                self._comm_stream = torch.cuda.Stream(device=dev)
            comm, main = self._comm_stream, torch.cuda.current_stream(dev)
            pool = self.__dict__.setdefault("_ex_events", {})
            pair = pool.get(self.collective_count)          # one event pair per message of the step, made once
            if pair is None:
                pair = pool[self.collective_count] = (torch.cuda.Event(), torch.cuda.Event())
            e0, ev = pair
            e0.record(main)
            comm.wait_event(e0)
            _lib.check(_lib.lib().sp_stream_delay_us(self.sync_bn_latency_us, _lib.c_void_p(comm.cuda_stream)), "delay")
            ev.record(comm)
        return work, ev

    def _exchange_wait(self, token) -> None:
        work, ev = token
        if work is not None:
            work.wait()                       # RCCL: the compute stream waits for the collective's stream; gloo: the host does
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _wgrad_flush(self) -> None:
        """Launch the queued weight gradients as one group (on the wgrad stream when `overlap_wgrad`)."""
        q = getattr(self, "_wg_queue", None)
        if not q:
            return
        lib, B, side, dev = _lib.lib(), self._wg_batch, self._wg_side, self._wg_dev
        jobs = (_lib.WgradJob * len(q))()
        for j, (layer, xin, dzt) in zip(jobs, q):
            layer.wgrad_job(xin, dzt, B, j)
        need = ctypes.c_int64(0)
        _lib.check(lib.sp_conv2d_wgrad_workspace(jobs, len(q), ctypes.byref(need)), "wgrad workspace")
        if need.value > self.wgrad_ws.numel() * 4:
            if side is not None:
                side.synchronize()                      # (first steps only: the slab buffer grows to the largest group)
            self.wgrad_ws = torch.empty((need.value + 3) // 4 + (1 << 20), dtype=torch.float32, device=dev)
        flops = sum(layer.flops for layer, _, _ in q)
        ke = self.kernel_events
        if side is None:
            st = torch.cuda.current_stream(dev)
            if ke is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
            _lib.check(lib.sp_conv2d_wgrad_batched(jobs, len(q), P(self.wgrad_ws), self.wgrad_ws.numel() * 4, _lib.current_stream()), "wgrad group")
            if ke is not None:
                e1.record(st)
                ke.append(("wgrad", f"group of {len(q)}", flops, e0, e1))
        else:
            main = torch.cuda.current_stream(dev)
            n = self._wg_flushes = getattr(self, "_wg_flushes", 0) + 1
            ev = self._wgrad_events.get(n)
            if ev is None:
                ev = self._wgrad_events[n] = (torch.cuda.Event(), torch.cuda.Event())
            ev[0].record(main)                              # every queued dz (and everything before it) is ready
            side.wait_event(ev[0])
            if ke is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
            _lib.check(lib.sp_conv2d_wgrad_batched(jobs, len(q), P(self.wgrad_ws), self.wgrad_ws.numel() * 4, _lib.c_void_p(side.cuda_stream)),
                       "wgrad group")
            if ke is not None:
                e1.record(side)
                ke.append(("wgrad", f"group of {len(q)}", flops, e0, e1))
            for _, _, dzt in q:
                dzt.record_stream(side)                     # dz was released by the tape already: the allocator must wait for `side`
            ev[1].record(side)
            self._wgrad_tail = ev[1]
        self._wg_queue = []
        self._wg_queued_flops = 0.0

    def _wgrad_flush_if(self, fraction: float) -> None:
        if getattr(self, "_wg_queued_flops", 0.0) >= fraction * getattr(self, "_group_gflop", self.wgrad_group_gflop) * 1e9:
            self._wgrad_flush()

    wgrad_group_gflop = 40.0   # queued weight-gradient work (all images of the rank) that triggers a group launch

    # ---- static structure -------------------------------------------------------------------------------------------
    def _conv(self, name, h, w, **kw) -> ConvT:
        layer = ConvT(self, name, kw.pop("kind", "conv"), self.sd[name + ".weight"].detach(), h, w, **kw)
        self.layers[name] = layer
        return layer

    def _build_hrnet(self, H, W):
        """One ConvT per conv of PoseHighResolutionNet (nets/pose_hrnet.py:419-454), walked exactly as the forward walks it."""
        extra = self.model.cfg["MODEL"]["EXTRA"]
        cp = 8 if self.bf16 else 4
        self._conv("conv1", H, W, stride=2, pad=1, c_in_buf=cp, need_dgrad=False, taps_w=8)      # K = 3 x 8 x cp: whole 128-byte tiles
        self._conv("conv2", H // 2, W // 2, stride=2, pad=1)
        h, w = H // 4, W // 4
        for k in range(4):
            p = f"layer1.{k}"
            self._conv(p + ".conv1", h, w)
            self._conv(p + ".conv2", h, w, pad=1)
            self._conv(p + ".conv3", h, w)
            if k == 0:
                self._conv(p + ".downsample.0", h, w)
        pre_n = 1
        for si, st in enumerate((2, 3, 4)):
            sc = extra[f"STAGE{st}"]
            nb = sc["NUM_BRANCHES"]
            t = f"transition{si + 1}"
            for i in range(nb):
                if i < pre_n:
                    if (f"{t}.{i}.0.weight") in self.sd:
                        self._conv(f"{t}.{i}.0", h >> i, w >> i, pad=1)
                else:
                    for j in range(i + 1 - pre_n):
                        self._conv(f"{t}.{i}.{j}.0", h >> (pre_n - 1 + j), w >> (pre_n - 1 + j), stride=2, pad=1)
            for m in range(sc["NUM_MODULES"]):
                multi = not (st == 4 and m == sc["NUM_MODULES"] - 1)
                base = f"stage{st}.{m}"
                for i in range(nb):
                    for k in range(sc["NUM_BLOCKS"][i]):
                        self._conv(f"{base}.branches.{i}.{k}.conv1", h >> i, w >> i, pad=1)
                        self._conv(f"{base}.branches.{i}.{k}.conv2", h >> i, w >> i, pad=1)
                for i in range(nb if multi else 1):
                    for j in range(nb):
                        f = f"{base}.fuse_layers.{i}.{j}"
                        if j > i:
                            self._conv(f + ".0", h >> j, w >> j)
                        elif j < i:
                            for k in range(i - j):
                                self._conv(f"{f}.{k}.0", h >> (j + k), w >> (j + k), stride=2, pad=1)
            pre_n = nb
        kf = extra["FINAL_CONV_KERNEL"]
        self._conv("final_layer", h, w, pad=1 if kf == 3 else 0, bias_name="final_layer.bias", out_nchw=True)
        self.heat_hw = (h, w)

    def _build(self, H, W):
        if self.head == "hrnet":
            return self._build_hrnet(H, W)
        self._conv("conv1", H, W, stride=2, pad=3, c_in_buf=8 if self.bf16 else 4, need_dgrad=False)
        h, w = H // 4, W // 4
        inpl = 64
        basic = getattr(self.model, "BLOCK", "bottleneck") == "basic"
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), self.model.BLOCKS), start=1):
            for bi in range(n):
                s = 2 if (bi == 0 and li > 1) else 1
                p = f"layer{li}.{bi}"
                if basic:
                    # BasicBlock (resnet18 / resnet34, pose_resnet_dconv.py:38-80): conv1 3x3 carries the stride, conv2 3x3; a projection
                    # shortcut (and, with reduction=True, the SELayer) only where the shape changes
                    self._conv(p + ".conv1", h, w, stride=s, pad=1)
                    self._conv(p + ".conv2", h // s, w // s, pad=1)
                    if (p + ".downsample.0.weight") in self.sd:
                        self._conv(p + ".downsample.0", h, w, stride=s)
                    if (p + ".se.fc.0.weight") in self.sd:
                        self._conv(p + ".se.fc.0", 1, 1)
                        self._conv(p + ".se.fc.2", 1, 1)
                    h, w = h // s, w // s
                    inpl = planes
                    continue
                self._conv(p + ".conv1", h, w)
                self._conv(p + ".conv2", h, w, stride=s, pad=1, groups=getattr(self.model, "GROUPS", 1))      # (resnext*: groups = 32, :97-101)
                self._conv(p + ".conv3", h // s, w // s)
                if bi == 0:
                    self._conv(p + ".downsample.0", h, w, stride=s)
                if (p + ".se.fc.0.weight") in self.sd:          # SELayer (reduction=True): its two 1x1 convs act on the pooled [B,1,1,C] map
                    self._conv(p + ".se.fc.0", 1, 1)
                    self._conv(p + ".se.fc.2", 1, 1)
                h, w = h // s, w // s
                inpl = planes * 4
        if self.head == "dconv":
            for idx in (0, 3, 6):
                self._conv(f"deconv_layers.{idx}", h, w, kind="deconv", stride=2, pad=1)
                h, w = 2 * h, 2 * w
            self._conv("final_layer", h, w, bias_name="final_layer.bias", out_nchw=True)
        else:                                              # PixelShuffle, DUC(512->1024), DUC(256->512), conv3x3 (pose_resnet_duc.py:227-232)
            h, w = 2 * h, 2 * w
            for idx in (1, 2):
                self._conv(f"duc_layers.{idx}.conv", h, w, pad=1)
                h, w = 2 * h, 2 * w
            self._conv("final_layer", h, w, pad=1, bias_name="final_layer.bias", out_nchw=True)
        self.heat_hw = (h, w)

    def repack(self, rows_range=None, stream=None, bucket: Optional[int] = None):
        """Regenerate every packed weight copy from the (just updated) flat parameter buffer: one launch over a device-side
        job table (141 pack jobs for ResNet50-DConv, cut into ~1,200 equal-sized slabs so that the grid is balanced)."""
        if getattr(self, "_pack_table", None) is None:
            import numpy as np
            rec = np.dtype([("d", "<i4", 4), ("s", "<i8", 4), ("lim", "<i4", 4), ("base", "<i8"), ("dst", "<i8"), ("total", "<i8"),
                            ("bf16", "<i4"), ("pad", "<i4"), ("tap0", "<i4"), ("tile0", "<i4")], align=True)
            jobs = [j for layer in self.layers.values() for j in layer.pack_jobs]
            rows = []
            for j in jobs:                                     # cut every job into slabs of <= ~64 K elements along its first axis
                o, _ = self.flat.offsets[j.src_name]
                es = 2 if j.dst.dtype == torch.bfloat16 else 4
                inner = int(np.prod(j.dims[1:]))
                step = max(1, 65536 // inner)
                for r0 in range(0, j.dims[0], step):
                    n0 = min(step, j.dims[0] - r0)
                    # how the kernel walks the slab (sp_permute4_batched): filters with taps read a run of taps per (i0, i3) pair, 1x1
                    # filters packed transposed go through LDS tiles, the rest (sources contiguous along the last index) in destination order
                    walk = 1 if j.dims[1] * j.dims[2] > 1 else (2 if abs(j.strides[0]) == 1 and abs(j.strides[3]) > 1 else 0)
                    tap0 = tile0 = 0
                    if walk == 1 and self.repack_tiled and 0 < j.strides[0] < abs(j.strides[3]) and 32 * (j.strides[0] | 1) <= 10240 and j.dims[1] * j.dims[2] <= 32:
                        # the fastest destination index is the source's slowest (dgrad packs of k > 1 convs, a transposed conv's phase packs):
                        # tiles through LDS, coalesced both ways (sp_permute4_batched_tiled walk 3).  `j.base` is the tap offset inside an
                        # (i0, i3) pair's block of strides[0] source floats (9 for a 3x3 filter, 16 for a transposed conv's 4x4: the gate
                        # above admits any block of which at least ONE i0 row per i3 fits the 40 KB tile; tile0 = how many do).
                        walk, tap0 = 3, -int(j.base)
                        tile0 = 32
                        while 32 * ((tile0 * j.strides[0]) | 1) > 10240:
                            tile0 //= 2
                        assert tile0 >= 1
                    rows.append(((n0,) + tuple(j.dims[1:]), j.strides, (max(0, min(n0, j.valid[0] - r0)),) + tuple(j.valid[1:]),
                                 o + j.base + r0 * j.strides[0], j.dst.data_ptr() + (j.dst_off + r0 * inner) * es, n0 * inner,
                                 int(j.dst.dtype == torch.bfloat16), walk, tap0, tile0))
            tab = np.zeros(len(rows), dtype=rec)
            for i, (d, st, lim, base, dst, total, b16, walk, tap0, tile0) in enumerate(rows):
                tab[i]["d"], tab[i]["s"], tab[i]["lim"] = d, st, lim
                tab[i]["base"], tab[i]["dst"], tab[i]["total"], tab[i]["bf16"], tab[i]["pad"] = base, dst, total, b16, walk
                tab[i]["tap0"], tab[i]["tile0"] = tap0, tile0
            assert rec.itemsize == 104, rec.itemsize
            self._pack_table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(self.flat.data.device)
            self._pack_n = len(rows)
            self._pack_tiled = any(r[7] == 3 for r in rows)    # walk 3 needs the launch that owns the 40 KB LDS tile (sp_permute4_batched_tiled)
            # rows are in parameter order: the rows that read from a gradient bucket's slice of the flat buffer are one range
            self._pack_rows_of_bucket = []
            src_off = [int(r[3]) for r in rows]
            for b in self.buckets:
                idx = [i for i, o in enumerate(src_off) if b["lo"] <= o < b["hi"]]
                assert not idx or idx == list(range(idx[0], idx[-1] + 1)), "pack rows of a bucket must be contiguous"
                self._pack_rows_of_bucket.append((idx[0], idx[-1] + 1) if idx else (0, 0))
        lo, hi = (0, self._pack_n) if rows_range is None else rows_range
        if rows_range is None:
            self._packed_version = self._param_version()
        if hi > lo:
            tab = _lib.c_void_p(self._pack_table.data_ptr() + 104 * lo)
            launch = _lib.lib().sp_permute4_batched_tiled if self._pack_tiled else _lib.lib().sp_permute4_batched
            _lib.check(launch(P(self.flat.data), tab, hi - lo, 8, stream if stream is not None else _lib.current_stream()), "repack")
        # grouped layers (resnext*): their block-diagonal panels are not an affine gather - one small launch per packed copy
        for layer in self._grouped_layers:
            if bucket is None or self._bucket_of[layer.wname] == bucket:
                layer.pack_grouped(stream if stream is not None else _lib.current_stream())

    # ---- per-layer tile choice -----------------------------------------------------------------------------------------------
    def autotune(self, batch: int, reps: int = 5, rounds: int = 3) -> Dict[str, tuple]:
        """Time every legal implicit-GEMM tile of every forward and dgrad launch at this per-GPU batch (HIP events, random operands of
        the launch's real shapes) and pin the fastest in the descriptors (`tile_m` / `tile_n`).  Conv outputs and gradients do not depend
        on the tile (same reduction order); the BatchNorm partial sums are grouped per tile row block (fp32 sums of <= 64 values,
        folded in fp64), so batch statistics agree across tile tables to fp32 rounding (~1e-7 relative), not bit for bit (gradients
        then differ by what this net makes of such a rounding: up to 4e-4 relative L2 in fp32, far more in bf16).  The
        built-in heuristic was fitted at bs=128 inference shapes.  The STATS / BSTATS
        epilogues ride on the same tiles, so the plain launch is timed as their stand-in.  Untimed setup: call once before training."""
        lib, stream, dev = _lib.lib(), _lib.current_stream(), self.flat.data.device
        chosen: Dict[tuple, tuple] = {}
        report: Dict[str, tuple] = {}

        def best_tile(d, w) -> tuple:
            d.batch = batch
            key = tuple(getattr(d, f) for f, _ in ConvDesc._fields_ if f not in ("tile_m", "tile_n", "kernel"))
            if key in chosen:
                return chosen[key]
            bf = bool(d.flags & SP_CONV_BF16)
            x = torch.randn((batch, d.in_h, d.in_w, d.c_in), device=dev).to(torch.bfloat16 if bf else torch.float32)
            out_f32 = (not bf) or bool(d.flags & (SP_CONV_OUT_F32 | SP_CONV_OUT_NCHW))
            y = torch.empty((batch, d.out_h, d.out_w, max(d.out_c, 1)), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
            timed = []
            for bm, bn in _lib.CONV_TILES:
                if d.n_pad % bn:
                    continue
                d.tile_m, d.tile_n = bm, bn
                if lib.sp_conv2d_fwd(d, P(x), P(w), None, None, None, P(y), stream) != 0:
                    continue                                    # a tile this descriptor cannot use
                ts = []
                for _ in range(rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        lib.sp_conv2d_fwd(d, P(x), P(w), None, None, None, P(y), stream)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / reps)
                timed.append((sorted(ts)[rounds // 2], (bm, bn)))
            assert timed, "no legal tile"
            chosen[key] = min(timed)[1]
            return chosen[key]

        for name, layer in self.layers.items():
            t = best_tile(layer.d_fwd, layer.w_fwd)
            layer.d_fwd.tile_m, layer.d_fwd.tile_n = t
            report[name] = t
            if layer.need_dgrad:
                for i, (d, w) in enumerate(zip(layer.d_dgrad, layer.w_dgrad)):
                    t = best_tile(d, w)
                    d.tile_m, d.tile_n = t
                    report[f"{name}.dgrad{i}"] = t
        self.tuned_for_batch = batch
        return report

    tuned_for_batch = 0

    def get_tiles(self) -> Dict[str, list]:
        """The pinned tile of every forward / dgrad launch ("<layer>" / "<layer>.dgrad<i>" -> [tile_m, tile_n]; [0, 0] = built-in choice)."""
        out = {}
        for name, layer in self.layers.items():
            out[name] = [layer.d_fwd.tile_m, layer.d_fwd.tile_n]
            if layer.need_dgrad:
                for i, d in enumerate(layer.d_dgrad):
                    out[f"{name}.dgrad{i}"] = [d.tile_m, d.tile_n]
        return out

    def set_tiles(self, table: Dict[str, list], batch: int) -> None:
        """Pin a tile table (`get_tiles` / `autotune` of another run or rank).  The BatchNorm partial sums are grouped per tile row block,
        so ranks (and runs) that must agree to the last bit share ONE table: `autotune_shared` tunes on rank 0 and broadcasts it."""
        def pin(d, t):
            tm, tn = (int(v) for v in t)
            if d.c_in_group and tn != d.c_in_group:
                return                                   # a grouped launch's N tile IS its weight panel: an entry of another backbone (same layer names) is not for it
            d.tile_m, d.tile_n = tm, tn
        for name, layer in self.layers.items():
            if name in table:
                pin(layer.d_fwd, table[name])
            if layer.need_dgrad:
                for i, d in enumerate(layer.d_dgrad):
                    if f"{name}.dgrad{i}" in table:
                        pin(d, table[f"{name}.dgrad{i}"])
        self.tuned_for_batch = batch

    def autotune_shared(self, batch: int) -> Dict[str, list]:
        """`autotune` on rank 0, the table broadcast to every rank: tile choice is a timing outcome, and per-rank choices would make the
        BatchNorm statistics (hence the whole step) differ between ranks and between runs by rounding."""
        import torch.distributed as dist
        if self.world == 1:
            self.autotune(batch)
            return self.get_tiles()
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
        box = [None]
        if dist.get_rank() == src:
            self.autotune(batch)
            box[0] = self.get_tiles()
        dist.broadcast_object_list(box, src=src, group=self.pg)
        self.set_tiles(box[0], batch)
        return box[0]

    # ---- keeping the packed copies in step with the parameters -------------------------------------------------------------
    def _param_version(self) -> int:
        return sum(p._version for p in self.sd.values())

    def refresh_packed_weights(self) -> None:
        """Repack if a parameter changed behind the trainer's back (a torch optimizer's step, load_state_dict: in-place writes that
        bump the tensors' version counters).  The trainer's own Adam kernel + repack do not, and keep the copies current themselves."""
        if self._param_version() != getattr(self, "_packed_version", None):
            self.repack()

    def still_owns_parameters(self) -> bool:
        """False once the module's parameters no longer live in this trainer's flat buffer (model.to(...), a re-assigned .data)."""
        base, n = self.flat.data.data_ptr(), self.flat.numel * 4
        return all(base <= p.data_ptr() < base + n for p in self.sd.values())

    # ---- autograd surface (nets.*.forward in train mode) -----------------------------------------------------------------------
    def autograd_backward(self, backward, dheat: torch.Tensor, params) -> tuple:
        """Run the recorded tape and hand the parameter gradients to autograd as views of the flat gradient buffer (autograd then
        sets / accumulates `p.grad`, fires hooks - DDP's reducer included).  The kernels OVERWRITE their gradient buffer, so when
        live `.grad` tensors still alias it (zero_grad(set_to_none=False), gradient accumulation) the tape writes into a second
        flat buffer and autograd's `p.grad += returned` is then the correct accumulation."""
        def aliased(buf):
            lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * 4
            return any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in params)
        clone = False
        if aliased(self.flat.grad):
            if getattr(self, "_grad_alt", None) is None:
                self._grad_alt = torch.zeros_like(self.flat.grad)
            self.flat.grad, self._grad_alt = self._grad_alt, self.flat.grad
            clone = aliased(self.flat.grad)            # both buffers referenced by live gradients: return copies
        backward(dheat)
        out = []
        for name, p in self.sd.items():
            g = self.flat.view(name, grad=True).view(p.shape)
            out.append(g.clone() if clone else g)
        return tuple(out)

    # ---- one step -----------------------------------------------------------------------------------------------------
    def forward_backward(self, x: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """x [B,3,H,W], targets [B,J,H/4,W/4], mask [B,J] on the GPU -> loss (device scalar); gradients land in the flat
        gradient buffer (every element is overwritten, no zero_grad needed)."""
        lib, stream = _lib.lib(), _lib.current_stream()
        targets = _lib.require_cuda_f32(targets, "targets")
        mask = _lib.require_cuda_f32(mask, "mask")
        if getattr(self, "_arena", None) is None:
            self._arena = []
        self._arena_on, self._arena_i = self.use_arena, 0
        try:
            heat, backward = self.forward_tape(x)
            B, J, hh, ww = heat.shape
            # ---- loss + d loss / d heat ----
            dheat = self._take(tuple(heat.shape), torch.float32, heat.device)
            _lib.check(lib.sp_masked_mse(P(heat), P(targets), P(mask), B, J, hh * ww, P(self.loss_buf), P(dheat), P(self.mse_ws), stream), "mse")
            self._mark("forward_loss")
            backward(dheat)
        finally:
            self._arena_on = False
        return self.loss_buf

    def forward_tape(self, x: torch.Tensor):
        """See `_forward_tape`.  While the tape is being issued the handle of the stream it goes to is pinned (`_lib.pin_stream`): the ~170
        `torch.cuda.current_stream()` lookups of a step were a sixth of its host time."""
        prev = _lib.pin_stream((_lib.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), _lib._device_index(x.device)))
        try:
            heat, backward = self._forward_tape(x)
        finally:
            _lib.pin_stream(prev)

        def pinned_backward(dheat: torch.Tensor) -> None:
            prev = _lib.pin_stream((_lib.c_void_p(torch.cuda.current_stream(dheat.device).cuda_stream), _lib._device_index(dheat.device)))
            try:
                backward(dheat)
            finally:
                _lib.pin_stream(prev)
        return heat, pinned_backward

    def _forward_tape(self, x: torch.Tensor):
        """Train-mode forward (batch-statistics BatchNorm, running statistics updated): x [B,3,H,W] -> (heat maps [B,J,H/4,W/4],
        backward) where `backward(dheat)` runs the recorded tape - dgrad / wgrad / BN backward - and leaves every parameter gradient
        in the flat gradient buffer `self.flat.grad` (overwritten, not accumulated).  `forward_backward` = this + the masked-MSE
        kernel; `nets.*.forward` in train mode = this behind a torch.autograd.Function (reference loop ddp...:114-119)."""
        lib, stream = _lib.lib(), _lib.current_stream()
        x = _lib.require_cuda_f32(x, "input")
        B, dev = x.shape[0], x.device
        self.refresh_packed_weights()
        if getattr(self.model, "_program", None) is not None:
            self.model._program = None        # running statistics change below: the eval-mode program folds them into its weights
        # the forward launches and their backward closures: simple_pose_amd/tape.py (primitives + one builder per net family)
        from . import tape as tape_mod
        a, t = tape_mod.record(self, x)
        return self._finish_forward(a, t.tape, t.nbt, B, t.wgrad_async, t.new, t.newf)

    def _finish_forward(self, a: Act, tape, nbt, B, wgrad_async, new, newf):
        """final_layer (+ bias, NCHW heat maps) on the last activation; returns (heat maps, backward closure)."""
        fl = self.layers["final_layer"]
        J = fl.O
        hh, ww = self.heat_hw
        heat = torch.empty((B, J, hh, ww), dtype=torch.float32, device=a.data.device)
        a.consumers += 1
        fl.forward(a.data, B, out=heat, shift=self.sd["final_layer.bias"])
        self.last_heat = heat
        torch._foreach_add_(nbt, 1)                        # num_batches_tracked of every BatchNorm (train-mode forward)
        last = a

        def backward(dheat: torch.Tensor) -> None:
            self._backward_tape(tape, last, fl, dheat, B, J, hh, ww, wgrad_async, new, newf)
        return heat, backward

    def _backward_tape(self, tape, a, fl, dheat, B, J, hh, ww, wgrad_async, new, newf) -> None:
        lib, stream = _lib.lib(), _lib.current_stream()
        dheat = _lib.require_cuda_f32(dheat, "d loss / d heat maps")
        dev, ws = dheat.device, self.red_ws
        if True:   # (block kept for a minimal diff of the tape below)
            Jb = fl.c_out_buf                              # heat-map channels padded to a K tile of the backward launches
            # final_layer.bias.grad = sum over batch and pixels of d loss / d heat, straight from the NCHW gradient into the flat buffer
            _lib.check(lib.sp_channel_sum_nchw(P(dheat), B, J, hh * ww, P(self.flat.view("final_layer.bias", True)), stream), "final_layer.bias.grad")
            dh = new((B, hh, ww, Jb)) if self.bf16 else newf((B, hh, ww, Jb))
            _lib.check(lib.sp_nchw_to_nhwc_pad(P(dheat), P(dh), int(self.bf16), B, J, hh, ww, Jb, stream), "dheat.nhwc")
            wgrad_async(fl, a.data, dh)
            a.grad = fl.dgrad(dh, B, None, bn_src=a if (self.fuse_bn_bwd and a.bn is not None and a.consumers == 1) else None)
            a.contrib += 1
            self._grads_ready("final_layer.bias", "final_layer.weight")
            for fn in reversed(tape):
                fn()
            for xa in list(getattr(self, "_branch_open", [])):       # (a branch nobody downstream joined: its jobs still have to be queued)
                self._join_grad(xa)
            self._wgrad_flush()
            if self._wgrad_tail is not None:
                torch.cuda.current_stream(dev).wait_event(self._wgrad_tail)      # join: the optimizer reads every weight gradient
            if self._opt_in_backward:
                assert all(w is not None for w in self._works), "a gradient bucket never completed"
                for w in self._works:
                    torch.cuda.current_stream(dev).wait_event(w)                 # join: parameters and packed copies are updated
                self._works = None
        self._mark("backward")

    def all_reduce_grads(self) -> float:
        """DDP semantics: gradients are averaged over ranks.  SUM all-reduces of the flat buffer's buckets (RCCL over xGMI) were
        launched during backward; wait for them here (launch whatever backward did not cover).  The 1/world factor is folded
        into the Adam kernel."""
        import torch.distributed as dist

        if self.world == 1:
            return 1.0
        works = getattr(self, "_works", None)
        if works is None:                                  # gradients filled by hand (tests): one all-reduce of everything
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.pg)
            return 1.0 / self.world
        for i, b in enumerate(self.buckets):
            if works[i] is None:
                works[i] = dist.all_reduce(self.flat.grad[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        for w in works:
            w.wait()
        self._works = None
        return 1.0 / self.world

    def _adam(self, sl: slice, grad_scale: float, stream) -> None:
        """torch.optim.Adam on a slice of the flat buffers (ddp...:70-72,119).  Under `capture()` the step's scalars live in device memory
        (written once per step by `_adam_scalars_for_step`), so the launch is identical every step."""
        lib, n = _lib.lib(), sl.stop - sl.start
        if self._adam_scalars is not None:
            _lib.check(lib.sp_adam_step_dev(P(self.flat.data[sl]), P(self.flat.grad[sl]), P(self.exp_avg[sl]), P(self.exp_avg_sq[sl]), n,
                                            P(self._adam_scalars), stream), "adam")
        else:
            _lib.check(lib.sp_adam_step(P(self.flat.data[sl]), P(self.flat.grad[sl]), P(self.exp_avg[sl]), P(self.exp_avg_sq[sl]), n,
                                        self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, grad_scale, stream), "adam")

    def _adam_scalars_for_step(self, grad_scale: float) -> None:
        """Next optimizer step: count it and hand its scalars to the device (outside the captured region, before the replay)."""
        self.step_count += 1
        _lib.check(_lib.lib().sp_adam_set_scalars(self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, grad_scale,
                                                  P(self._adam_scalars), _lib.current_stream()), "adam scalars")

    def optimizer_step(self, grad_scale: float = 1.0):
        if self._adam_scalars is not None:
            self._adam_scalars_for_step(grad_scale)
        else:
            self.step_count += 1
        self._adam(slice(0, self.flat.numel), grad_scale, _lib.current_stream())
        self.repack()
        if getattr(self.model, "_program", None) is not None:
            self.model._program = None        # the eval-mode program holds BN-folded copies of the weights the kernel just changed

    def step(self, x, targets, mask) -> torch.Tensor:
        """optimizer.zero_grad(); loss = ...; loss.backward(); optimizer.step()  (ddp...:114-119).  The optimizer runs inside
        backward, bucket by bucket (see `_grads_ready`); `fuse_optimizer = False` gives the three separate phases.
        Tried in round 4 and dropped: the main chain on a HIGH-priority HIP stream (so that the weight-gradient / optimizer streams only get
        the CUs the chain leaves free at dispatch): 6.32 vs 6.30 ms in a same-box A/B - queue priority orders dispatch, it does not preempt
        the 50 us weight-gradient workgroups that are already resident when a chain kernel arrives."""
        self._mark("start")
        if not self.fuse_optimizer:
            loss = self.forward_backward(x, targets, mask)
            scale = self.all_reduce_grads()
            self._mark("allreduce_wait")
            self.optimizer_step(scale)
            self._mark("adam_repack")
            return loss
        if self._adam_scalars is None:
            self.step_count += 1
        elif not self._adam_scalars_external:              # device-side scalars (a GraphedStep exists) but this step runs eagerly
            self._adam_scalars_for_step(1.0 / self.world)  # (a captured / replayed step is counted by GraphedStep, outside the graph)
        self._opt_in_backward = True
        try:
            loss = self.forward_backward(x, targets, mask)
        finally:
            self._opt_in_backward = False
        self._mark("allreduce_wait")
        if getattr(self.model, "_program", None) is not None:
            self.model._program = None
        self._mark("adam_repack")
        return loss

    def capture(self, x: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor, warmup: int = 3) -> "GraphedStep":
        """Record `step()` for this batch shape into ONE hipGraph: the ~400 launches on three streams (and, with a process group, RCCL's
        collectives) become a single graph launch per step - the host needs ~4 ms to enqueue a 6.3 ms bf16 step kernel by kernel,
        and twice that once SyncBatchNorm adds its 104 messages.  See GraphedStep."""
        return GraphedStep(self, x, targets, mask, warmup)

    _adam_scalars = None       # device [8] floats while a captured step owns the optimizer's scalars (see capture())
    _adam_scalars_external = False   # True while GraphedStep captures: the scalars' launch then sits outside the recorded region
    force_collectives = False  # issue the SyncBatchNorm / gradient collectives on ONE rank too (a 1-rank RCCL group: preflight of their capture)
    fuse_optimizer = True
    use_arena = True           # step / forward_backward reuse last step's tensors request by request (see _take)
    kernel_events = None       # bench.py: a list collects (kind, layer, flops per image, start event, end event) per conv-family launch
    _opt_in_backward = False   # class-level defaults: also valid for partially constructed instances (host-logic tests)
    _opt_stream = None
    _wgrad_stream = None
    _wgrad_tail = None
    fuse_stem_pool = os.environ.get("SP_STEM_POOL", "1") != "0"        # (env: development knob)
    _branch_stream = None
    _branch_main = None
    _in_branch = False
    overlap_shortcut = True
    relu_bit_masks = os.environ.get("SP_RELU_MASK", "1") != "0"        # (env: development knob)
    # bn2 + ReLU inside conv3's staging pass (tape.conv_bn defer_apply_to; sp_conv2d_fwd_bn_stats_abn): built and bit-identical in round 5, and
    # measured SLOWER in the step (same-box A/B x4: 5.80-5.84 ms against 5.75-5.78 - the map is re-applied once per N tile of the consumer and sits
    # in the MFMA waves' issue stream; the 16 launches it removes cost less).  Off; SP_ABN=1 switches it on.
    apply_in_consumer = os.environ.get("SP_ABN", "0") == "1"
    stem_bucket_params = int(os.environ.get("SP_STEM_BUCKET", "3"))      # (env: development knob; 0 = the stem shares layer1's bucket, rounds 1-4)
    repack_tiled = os.environ.get("SP_REPACK_TILED", "0") == "1"       # (walk 3 of sp_permute4_batched: fewer bytes, same-box A/B 5.83 vs 5.90 ms - off)
    lazy_residual_grad = os.environ.get("SP_LAZY_RES", "1") != "0"      # (env: development knob)
    fold_in_consumer_rows = 50    # partial rows (one per phase and M tile) up to which the BatchNorm pass folds them itself (sp_bn_fold_*): above, a stand-alone
                                  # fold + the plain pass (measured with the 16-byte bf16 passes: 1536 -> 6.06 ms, 100 -> 5.98, 50 -> 5.95, 0 -> 5.97)
    fuse_sync_finalize = True  # SyncBatchNorm: finalise inside the consuming bn_apply, message assembled by the backward fold (one launch less each way)
    fuse_bn_bwd = True         # BN backward sums from the epilogue of the dgrad launch that produces dy (single-consumer BN+ReLU outputs)
    fuse_bn_stats = True       # BN batch statistics from the conv epilogue (sp_conv2d_fwd_bn_stats), SyncBN included (sp_bn_sums_from_conv)
    collective_count = 0

    # ---- step-time split (BASELINE config 4 asks for fwd / bwd / all-reduce / Adam) -------------------------------------
    profile = False

    def _mark(self, name: str):
        if self.profile:
            if name == "start":
                self._marks = []
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    def phase_ms(self) -> Dict[str, float]:
        """Milliseconds of the last profiled step by phase (events on the compute stream; `allreduce_wait` is only the part of the
        gradient exchange that backward did not hide)."""
        torch.cuda.synchronize()
        m = self._marks
        return {m[i][0]: m[i - 1][1].elapsed_time(m[i][1]) for i in range(1, len(m))}


class GraphedStep:
    """`PoseTrainer.step` of one batch shape as a hipGraph (HIP stream capture through torch.cuda.CUDAGraph).

    What makes the step capturable: every tensor it touches has a fixed address (the step arena, the flat parameter / gradient / moment
    buffers, the packed weights, static copies of the inputs); the three streams of the step fork from and join the capturing stream
    through events; nothing in it synchronises with the host; and the only launch arguments that change from step to step - Adam's
    bias-corrected step size - are read from device memory (`sp_adam_step_dev`), refreshed by one tiny launch in front of every replay.
    With a process group the RCCL all-reduces (gradient buckets, SyncBatchNorm messages) are captured with the kernels.

        g = trainer.capture(x, targets, mask)      # a few eager steps (tile tables, arena, streams), then the capture
        loss = g.step(x, targets, mask)            # copies the batch into the static inputs, replays; same results as trainer.step

    The eager steps taken here are real optimizer steps on the given batch (as torch's own graph warm-up recipes do)."""

    def __init__(self, tr: PoseTrainer, x: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor, warmup: int = 3):
        if not tr.fuse_optimizer:
            raise ValueError("capture() records the fused step (optimizer inside backward)")
        if tr.profile or tr.kernel_events is not None:
            raise ValueError("capture(): timing events (profile / kernel_events) cannot be recorded inside a graph")
        self.tr = tr
        dev = tr.flat.data.device
        self.x = _lib.require_cuda_f32(x, "input").clone()
        self.targets = _lib.require_cuda_f32(targets, "targets").clone()
        self.mask = _lib.require_cuda_f32(mask, "mask").clone()
        tr._adam_scalars = torch.zeros(8, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0 / tr.world
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                 # eager, on the stream the capture will use: streams, events, arena, tile opt-ins
                tr.step(self.x, self.targets, self.mask)    # (counts itself and refreshes the device-side scalars)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        tr._adam_scalars_external = True
        try:
            with torch.cuda.graph(self.graph, stream=side):
                self.loss = tr.step(self.x, self.targets, self.mask)
        finally:
            tr._adam_scalars_external = False
        # the graph has the addresses of the step arena's tensors and of the weight-gradient slabs baked in: keep them alive whatever the
        # trainer does afterwards (an eager step at another batch size truncates the arena, _wgrad_flush may re-allocate the slabs)
        self._pinned = (list(tr._arena), tr.wgrad_ws)
        self.launches = None

    def step(self, x: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        if targets is not self.targets:
            self.targets.copy_(targets, non_blocking=True)
        if mask is not self.mask:
            self.mask.copy_(mask, non_blocking=True)
        self.tr._adam_scalars_for_step(self.grad_scale)
        self.graph.replay()
        return self.loss

    def release(self) -> None:
        """Give the optimizer's scalars back to the eager path (`trainer.step` works as before)."""
        self.tr._adam_scalars = None
        self.graph = None
        self._pinned = None
