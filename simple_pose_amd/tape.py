"""The train step's recorded forward ("tape"): primitives that launch the forward kernels of one op and append its backward closure,
and one builder per net family that strings them together exactly as the reference's `forward` does.

    StepTape(trainer, x)            state of ONE step: batch, streams, flag words, the tape (list of backward closures)
      .conv_bn / .se_gate / .upsample_add / .shuffle / .stem_fused / .stem_plain     primitives (forward launches + backward closure)
    build_hrnet / build_resnet_bottleneck / build_resnet_basic / build_head_dconv / build_head_duc     per-family builders
    record(trainer, x) -> (last activation, StepTape)                                what PoseTrainer._forward_tape calls

Until round 4 all of this was one 580-line method of closures (`PoseTrainer._forward_tape`); the launches, their order and every argument are
unchanged (the step is bit-identical: tests/test_gpu_train.py).  References: processors/ddp_pose_resnet_solver.py:110-133 (the step),
nets/pose_resnet_dconv.py:38-133,251-265 (BasicBlock / Bottleneck / forward), nets/pose_resnet_duc.py:227-251, nets/commons.py:4-43,
nets/pose_hrnet.py:181-259,419-454.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch

from . import _lib
from ._lib import ptr as P

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


class StepTape:
    """Forward of one train step on `tr` (a PoseTrainer): holds what the old closures shared."""

    def __init__(self, tr, x: torch.Tensor):
        from .train import Act  # noqa: F401  (type only; train imports this module lazily)
        self.tr = tr
        self.lib = _lib.lib()
        self.stream = _lib.current_stream()
        self.B, self.dev = x.shape[0], x.device
        self.tape: List[Callable[[], None]] = []
        self.nbt: List[torch.Tensor] = []
        self.L = tr.layers
        self.ws = tr.red_ws
        self.bf = int(tr.bf16)
        self.gf = self.bf | (2 if tr.g16 else 0)        # flag word of the backward passes: bit 0 = bf16 activations, bit 1 = bf16 activation gradients
        self.use_mask = tr.g16 and tr.relu_bit_masks     # (bit 2 of that word, per call: the ReLU source is the bit mask of Act.mask)
        self.sync, self.W = tr.sync_bn, tr.world
        dev = self.dev
        tr._wgrad_tail = None
        tr._wg_flushes = 0
        side = None
        if tr.overlap_wgrad:
            # weight gradients only feed the optimizer: they run on a second HIP stream beside the dgrad / BN-backward chain, which
            # at 32 images per GPU is a string of small launches that leave most CUs idle
            if tr._wgrad_stream is None:
                tr._wgrad_stream = torch.cuda.Stream(device=dev)
                tr._wgrad_events = {}
            side = tr._wgrad_stream
        tr._wg_queue = []
        tr._wg_queued_flops = 0.0
        tr._wg_batch, tr._wg_side, tr._wg_dev = self.B, side, dev
        # A stage's projection shortcut (conv + BatchNorm) depends on the block input alone: forward and backward it runs on a branch
        # stream beside conv1 -> bn1 -> conv2 -> bn2 -> conv3 (a chain of launches that each leave most of the chip idle).  Same
        # kernels, same accumulation order (the shortcut's share of the block input's gradient lands first, conv1's dgrad adds last
        # and after the join), so the bits do not change.  Off with SyncBatchNorm (the pair shares one message) and for the nets whose
        # shortcut passes use the shared reduction workspace.
        self.branch = None
        tr._in_branch = False
        tr._branch_open = []
        if (tr.overlap_shortcut and not tr.sync_bn and tr.fuse_bn_stats and tr.fuse_bn_bwd and tr.head != "hrnet"
                and getattr(tr.model, "BLOCK", "bottleneck") == "bottleneck" and not any(".se." in k for k in tr.layers)):
            if tr._branch_stream is None:
                tr._branch_stream = torch.cuda.Stream(device=dev)
                tr._branch_events = []
            self.branch = tr._branch_stream
        tr._branch_n = 0
        tr._join_grad = self.join_grad
        tr._group_gflop = float(os.environ.get("SP_WGRAD_GROUP_GFLOP", tr.wgrad_group_gflop))     # (env: development knob)
        tr._pending = [set(b["names"]) for b in tr.buckets]
        tr._works = [None] * len(tr.buckets)
        tr.collective_count = 0          # SyncBatchNorm all-reduces of this step (gradient buckets are counted in len(tr.buckets))

    # ---- tensors of the step (the trainer's arena) ------------------------------------------------------------------------------
    def new(self, shape, dtype=None):
        return self.tr._take((shape,) if isinstance(shape, int) else tuple(shape), dtype or self.tr.act_dtype, self.dev)

    def newf(self, shape):
        return self.tr._take((shape,) if isinstance(shape, int) else tuple(shape), torch.float32, self.dev)

    def newg(self, shape):
        return self.tr._take(tuple(shape), self.tr.grad_dtype, self.dev)

    # ---- branch stream -------------------------------------------------------------------------------------------------------------
    def branch_event(self):
        tr = self.tr
        n = tr._branch_n
        tr._branch_n = n + 1
        if n == len(tr._branch_events):
            tr._branch_events.append(torch.cuda.Event())
        return tr._branch_events[n]

    def run_on_branch(self, fn):
        """fn() with every launch on the branch stream, behind everything the current stream holds now; returns (result, event that
        marks the end of fn's launches on the branch stream)."""
        tr, branch, dev = self.tr, self.branch, self.dev
        here = torch.cuda.current_stream(dev)
        e0, e1 = self.branch_event(), self.branch_event()
        e0.record(here)
        branch.wait_event(e0)
        keep = self.stream
        tr._in_branch, tr._branch_main = True, here
        pinned = _lib.pin_stream((_lib.c_void_p(branch.cuda_stream), _lib._device_index(dev)))
        try:
            with torch.cuda.stream(branch):
                self.stream = _lib.current_stream()
                out = fn()
                e1.record(branch)
        finally:
            _lib.pin_stream(pinned)
            self.stream = keep
            tr._in_branch = False
        return out, e1

    def join_grad(self, xa) -> None:
        """Before .grad of `xa` is read or added to: wait for a branch that wrote it, then queue what the branch left for this stream."""
        if xa.grad_event is not None:
            torch.cuda.current_stream(self.dev).wait_event(xa.grad_event)
            xa.grad_event = None
            todo, xa.deferred = xa.deferred, []
            for f in todo:
                f()
            if xa in self.tr._branch_open:
                self.tr._branch_open.remove(xa)

    def wgrad_async(self, layer, xin: torch.Tensor, dzt: torch.Tensor):
        # weight gradients are launched in GROUPS (sp_conv2d_wgrad_batched: every layer of a group in one launch per dW tile shape plus
        # one fold launch): a layer alone has too few dW tiles to fill 256 CUs without cutting its pixels into hundreds of partial
        # slabs.  The group goes out when a gradient bucket completes, when `wgrad_group_gflop` of work is queued, or at the end.
        tr = self.tr
        if layer.groups > 1:
            # a grouped layer's dW is a streaming reduction of its own (sp_conv2d_wgrad_grouped), not a tile of the batched MFMA launch: it
            # goes out here, on the chain's stream, and writes its slice of the flat gradient buffer
            layer.wgrad_grouped(xin, dzt, self.B)
            return
        tr._wg_queue.append((layer, xin, dzt))
        tr._wg_queued_flops += layer.flops * self.B
        if tr._wg_queued_flops >= tr._group_gflop * 1e9 or len(tr._wg_queue) >= 64:     # (64 jobs: the batched call's limit)
            tr._wgrad_flush()

    # ---- conv + BatchNorm ------------------------------------------------------------------------------------------------------------
    def conv_stats(self, xa, cname: str) -> dict:
        """The conv launch of a conv + BatchNorm pair; with `fuse_bn_stats` its epilogue also leaves the per-channel partial sums."""
        layer = self.L[cname]
        xa.consumers += 1
        if xa.pending_apply is not None:
            # the input's BatchNorm + ReLU pass was deferred to this launch (conv_bn(defer_apply=True)): it forms the activation while it stages its A
            # operand and writes xa.data / xa.mask for the backward pass and the weight gradient
            zin, mean, invstd, gamma, beta = xa.pending_apply
            xa.pending_apply = None
            z, part, prow = layer.forward_bn_stats_abn(zin, self.B, mean, invstd, gamma, beta, xa.data, xa.mask)
        elif self.tr.fuse_bn_stats and layer.groups == 1:
            z, part, prow = layer.forward_bn_stats(xa.data, self.B)
        else:
            z, part, prow = layer.forward(xa.data, self.B), None, 0
        return dict(layer=layer, z=z, part=part, prow=prow, rows=z.shape[0] * z.shape[1] * z.shape[2], C=z.shape[3])

    def batch_stats(self, pends: List[dict], bnames: List[str], standalone: bool = False) -> None:
        """Batch mean / invstd (+ running statistics) of the BatchNorm layers behind the pending convs.  SyncBatchNorm: every
        layer folds its fp64 (sum, sum of squares) into a slice of ONE buffer and the group shares ONE all-reduce (the conv1 /
        downsample pair of a stage's first bottleneck); no second pass over z either way when the conv left partial sums."""
        tr, lib, stream, bf, ws, W = self.tr, self.lib, self.stream, self.bf, self.ws, self.W
        for pd in pends:
            pd["mean"], pd["invstd"] = self.newf(pd["C"]), self.newf(pd["C"])
        run = lambda bn: (P(tr.buffers[bn + ".running_mean"]), P(tr.buffers[bn + ".running_var"]))
        if not self.sync:
            for pd, bn in zip(pends, bnames):
                rm, rv = run(bn)
                if pd["part"] is not None and pd["prow"] <= tr.fold_in_consumer_rows and not standalone:
                    pd["fold_in_apply"] = True      # few partial rows: the consuming sp_bn_fold_apply_nhwc folds them in its prologue
                    continue
                if pd["part"] is not None:
                    part = pd["part"]
                    _lib.check(lib.sp_bn_train_stats_from_conv(P(part[0]), P(part[1]), pd["prow"], part.shape[2], pd["rows"], pd["C"], BN_EPS,
                                                               BN_MOMENTUM, P(pd["mean"]), P(pd["invstd"]), rm, rv, stream), bn)
                else:
                    _lib.check(lib.sp_bn_train_stats_nhwc(P(pd["z"]), bf, pd["rows"], pd["C"], BN_EPS, BN_MOMENTUM, P(pd["mean"]),
                                                          P(pd["invstd"]), rm, rv, P(ws), stream), bn)
            return
        sums = tr._take((2 * sum(pd["C"] for pd in pends),), torch.float64, self.dev)
        off = 0
        for pd, bn in zip(pends, bnames):
            pd["sums"] = sums[off:off + 2 * pd["C"]]
            off += 2 * pd["C"]
            if pd["part"] is not None:
                part = pd["part"]
                _lib.check(lib.sp_bn_sums_from_conv(P(part[0]), P(part[1]), pd["prow"], part.shape[2], pd["C"], P(pd["sums"]), stream), bn)
            else:
                _lib.check(lib.sp_bn_train_partial_nhwc(P(pd["z"]), bf, pd["rows"], pd["C"], P(pd["sums"]), P(ws), stream), bn)
        tr._exchange_wait(tr._exchange(sums))
        for pd, bn in zip(pends, bnames):
            if tr.fuse_sync_finalize:
                pd["from_sums"] = True         # the consuming sp_bn_apply_sums_nhwc finalises (mean, invstd, running statistics) itself
            else:
                rm, rv = run(bn)
                _lib.check(lib.sp_bn_train_finalize(P(pd["sums"]), pd["rows"] * W, pd["C"], BN_EPS, BN_MOMENTUM, P(pd["mean"]), P(pd["invstd"]),
                                                    rm, rv, stream), bn)

    def conv_bn(self, xa, cname: str, bname: str, relu: bool, res=None, pend: Optional[dict] = None,
                shortcut: bool = True, before_apply: Optional[Callable[[], None]] = None, defer_apply_to: Optional[str] = None):
        """conv -> train-mode BatchNorm (-> + res) (-> ReLU): forward launches, the resulting Act, and the backward closure on the tape.
        `defer_apply_to`: name of the ONE conv that consumes the result next - when that is a 1x1 convolution the kernels can feed from z
        (ConvT.abn_ok), the BatchNorm + ReLU pass is not launched: the consumer forms the activation in its staging pass and writes it."""
        from .train import Act
        tr, lib, bf, W, sync, B, dev = self.tr, self.lib, self.bf, self.W, self.sync, self.B, self.dev
        defer = (defer_apply_to is not None and tr.apply_in_consumer and relu and res is None and not sync and tr.fuse_bn_stats and pend is None
                 and before_apply is None and self.L[defer_apply_to].abn_ok(B) and self.L[defer_apply_to].ci == self.L[cname].O)
        if pend is None:
            pend = self.conv_stats(xa, cname)
            self.batch_stats([pend], [bname], standalone=defer)     # (deferred: the statistics come from the stand-alone fold, whatever the row count)
        if before_apply is not None:
            before_apply()                   # (the residual comes from the branch stream: join before the pass that reads it)
        stream = self.stream
        layer, z, mean, invstd = pend["layer"], pend["z"], pend["mean"], pend["invstd"]
        rows, C = pend["rows"], pend["C"]
        gamma, beta = tr.sd[bname + ".weight"], tr.sd[bname + ".bias"]
        self.nbt.append(tr.buffers[bname + ".num_batches_tracked"])
        y = self.new(z.shape)
        # bf16 gradients: the pass also leaves the ReLU mask as one bit per element; the BatchNorm backward pass and the dgrad epilogue
        # that reduces its sums then read that byte instead of 16 bytes of y
        mask = tr._take((rows * C // 8,), torch.uint8, dev) if (relu and self.use_mask and not pend.get("from_sums") and C % 8 == 0) else None
        if defer:
            pass                                 # y / mask are written by the consumer's launch (conv_stats sees pending_apply)
        elif pend.get("fold_in_apply"):
            part = pend["part"]
            _lib.check(lib.sp_bn_fold_apply_nhwc(P(z), bf, P(part[0]), P(part[1]), pend["prow"], part.shape[2], rows, BN_EPS, BN_MOMENTUM, P(gamma),
                                                 P(beta), P(res.data) if res else None, P(y), rows, C, int(relu), P(mean), P(invstd),
                                                 P(tr.buffers[bname + ".running_mean"]), P(tr.buffers[bname + ".running_var"]), P(mask),
                                                 stream), bname)
        elif pend.get("from_sums"):
            _lib.check(lib.sp_bn_apply_sums_nhwc(P(z), bf, P(pend["sums"]), rows * W, BN_EPS, BN_MOMENTUM, P(gamma), P(beta),
                                                 P(res.data) if res else None, P(y), rows, C, int(relu), P(mean), P(invstd),
                                                 P(tr.buffers[bname + ".running_mean"]), P(tr.buffers[bname + ".running_var"]), stream), bname)
        else:
            _lib.check(lib.sp_bn_apply_nhwc(P(z), bf, P(mean), P(invstd), P(gamma), P(beta), P(res.data) if res else None, P(y), rows, C,
                                            int(relu), P(mask), stream), bname)
        ya = Act(y, z.shape[1], z.shape[2], C)
        ya.mask = mask
        if defer:
            ya.pending_apply = (z, mean, invstd, gamma, beta)
        if res is not None:
            res.consumers += 1
        if relu:
            ya.bn = (z, mean, invstd)      # y = relu(bn(z) [+ res]): backward masks with y > 0 either way
        if not relu and res is None and shortcut:
            ya.sibling = (z, mean, invstd, bname)      # a projection shortcut: its backward sums ride on its consumer's (message / epilogue)
        if relu and res is not None and res.sibling is not None and tr.fuse_bn_bwd and (not sync or tr.fuse_sync_finalize):
            ya.bn2 = res.sibling                       # the shortcut's dy is this layer's g = dy * (y > 0): one more sum in the same epilogue

        def bwd():
            self._conv_bn_backward(xa, ya, res, layer, z, mean, invstd, y, gamma, rows, C, relu, cname, bname)
        self.tape.append(bwd)
        return ya

    def _conv_bn_backward(self, xa, ya, res, layer, z, mean, invstd, y, gamma, rows, C, relu, cname, bname) -> None:
        """Backward of one conv + BatchNorm (+ residual) (+ ReLU): the BatchNorm backward pass (its sums from wherever they were already
        reduced: the dgrad epilogue that completed dy, a SyncBatchNorm message, or its own reduction), the queued weight gradient, the
        conv's dgrad into the input's gradient."""
        tr, lib, W, sync, B, ws, gf = self.tr, self.lib, self.W, self.sync, self.B, self.ws, self.gf
        stream = self.stream                   # (read when the closure RUNS: a branch-stream section has switched it)
        new, newf, newg = self.new, self.newf, self.newg
        dz = new(z.shape)                      # MFMA operand of dgrad / wgrad: activation dtype
        dres = None
        acc = 0
        lazy = (res is not None and res.lazy_ok and res.grad is None and relu and ya.mask is not None and tr.g16
                and tr.lazy_residual_grad and tr.fuse_bn_bwd and not sync)
        if lazy:
            # the residual share g = dy * mask is not written: conv1's dgrad (the block input's other consumer, still to come on
            # the tape) adds it from (dy, mask) in its epilogue
            res.lazy_g = (ya.grad, ya.mask)
        elif res is not None:
            if res.grad is None:
                res.grad = newg(res.data.shape)   # activation gradients: fp32, or bf16 with grad_dtype "bf16"
            else:
                acc = 1
            dres = res.grad
        dgamma, dbeta = tr.flat.view(bname + ".weight", True), tr.flat.view(bname + ".bias", True)
        rs = P(y) if relu else None
        gm = gf                                # paths that take the bit mask: the two apply kernels (not the reduction passes)
        rsm = rs
        if relu and ya.mask is not None:
            rsm, gm = P(ya.mask), gf | 4
        if ya.presums is not None:
            # the consumer of this (projection-shortcut) BatchNorm already reduced (SyncBatchNorm: and exchanged) its two sums
            sg, sb = ya.presums
            ya.presums = None
            _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(ya.grad), gf, rs, P(z), P(mean), P(invstd), P(gamma), P(sg), P(sb),
                                                      rows * W, rows, C, P(dz), P(dres), acc, stream), bname + ".bwd")
        elif ya.bstats is not None and not sync and ya.bstats[1] <= tr.fold_in_consumer_rows:
            # few partial rows: ONE launch folds them (d beta, d gamma - and the projection shortcut's pair when its sum rode on the
            # same dgrad epilogue) in its prologue and applies the BatchNorm backward
            part, prow = ya.bstats
            ya.bstats = None
            three = part.shape[0] == 3 and res is not None and acc == 0
            dgs = dbs = None
            if three:
                sname = ya.bn2[3]
                dgs, dbs = tr.flat.view(sname + ".weight", True), tr.flat.view(sname + ".bias", True)
                res.presums = (dgs, dbs)
            _lib.check(lib.sp_bn_fold_bwd_apply_nhwc(P(ya.grad), gm, rsm, P(z), P(part[0]), P(part[1]), P(part[2]) if three else None, prow,
                                                     part.shape[2], P(mean), P(invstd), P(gamma), rows, rows, C, P(dgamma), P(dbeta), P(dgs),
                                                     P(dbs), P(dz), P(dres), acc, stream), bname + ".bwd")
        elif ya.bstats is not None or sync:
            if ya.bstats is not None:
                # the dgrad launch that completed ya.grad already reduced sum g and sum g*xhat (sp_conv2d_dgrad_bn_bwd_stats)
                part, prow = ya.bstats
                ya.bstats = None
                msg = None
                three = part.shape[0] == 3 and res is not None and acc == 0
                if sync and tr.fuse_sync_finalize and (three or not (res is not None and acc == 0 and res.sibling is not None)):
                    # SyncBatchNorm: the fold also writes the sums into the message (no concatenation launch); with the projection
                    # shortcut's sum g * xhat2 out of the same dgrad epilogue its pair travels in the same message
                    msg = newf((4 if three else 2) * C)
                    _lib.check(lib.sp_bn_bwd_sums_from_conv2(P(part[0]), P(part[1]), prow, part.shape[2], C, P(dgamma), P(dbeta),
                                                             P(msg[:C]), P(msg[C:2 * C]), stream), bname + ".bwd")
                    if three:
                        sname = ya.bn2[3]
                        dgs, dbs = tr.flat.view(sname + ".weight", True), tr.flat.view(sname + ".bias", True)
                        _lib.check(lib.sp_bn_bwd_sums_from_conv2(P(part[0]), P(part[2]), prow, part.shape[2], C, P(dgs), P(dbs),
                                                                 P(msg[2 * C:3 * C]), P(msg[3 * C:]), stream), sname + ".bwd")
                        res.presums = (msg[2 * C:3 * C], msg[3 * C:])       # (global after the exchange below)
                elif part.shape[0] == 3 and res is not None and acc == 0:
                    # the projection shortcut's sums came out of the same epilogue: d beta = sum g (shared), d gamma = sum g * xhat2;
                    # both BatchNorms' folds in one launch
                    sname = ya.bn2[3]
                    dgs, dbs = tr.flat.view(sname + ".weight", True), tr.flat.view(sname + ".bias", True)
                    _lib.check(lib.sp_bn_bwd_sums_from_conv_pair(P(part[0]), P(part[1]), P(part[2]), prow, part.shape[2], C, P(dgamma), P(dbeta),
                                                                 P(dgs), P(dbs), stream), bname + ".bwd")
                    res.presums = (dgs, dbs)
                else:
                    _lib.check(lib.sp_bn_bwd_sums_from_conv(P(part[0]), P(part[1]), prow, part.shape[2], C, P(dgamma), P(dbeta), stream),
                               bname + ".bwd")
            else:
                msg = None
                assert not tr._in_branch, "the shared reduction workspace is the main chain's"
                _lib.check(lib.sp_bn_train_bwd_reduce_nhwc(P(ya.grad), gf, rs, P(z), P(mean), P(invstd), rows, C, P(dgamma), P(dbeta),
                                                           P(ws), stream), bname + ".bwd")
            sg, sb, tot = dgamma, dbeta, rows
            if sync and msg is not None:
                token = tr._exchange(msg)
                tr._wgrad_flush_if(0.5)            # queued weight gradients go out under the message rather than after it
                tr._exchange_wait(token)
                sg, sb, tot = msg[:C], msg[C:2 * C], rows * W
            elif sync:
                # local sums are this rank's parameter gradients (DDP averages them later); dz needs the global ones
                parts = [dgamma, dbeta]
                sib = res.sibling if (res is not None and acc == 0) else None
                if sib is not None:
                    # the residual is a projection shortcut's BatchNorm output and this layer is its only consumer: its dy IS this
                    # layer's g = dy * (y > 0), so its two sums are reduced here and travel in the same message
                    zs, ms, ivs, sname = sib
                    dgs, dbs = tr.flat.view(sname + ".weight", True), tr.flat.view(sname + ".bias", True)
                    _lib.check(lib.sp_bn_train_bwd_reduce_nhwc(P(ya.grad), gf, rs, P(zs), P(ms), P(ivs), rows, C, P(dgs), P(dbs), P(ws),
                                                               stream), sname + ".bwd")
                    parts += [dgs, dbs]
                both = torch.cat(parts)
                token = tr._exchange(both)
                tr._wgrad_flush_if(0.5)            # queued weight gradients go out under the message rather than after it
                tr._exchange_wait(token)
                if sib is not None:
                    res.presums = (both[2 * C:3 * C], both[3 * C:])
                sg, sb, tot = both[:C], both[C:2 * C], rows * W
            _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(ya.grad), gm, rsm, P(z), P(mean), P(invstd), P(gamma), P(sg), P(sb), tot, rows, C,
                                                      P(dz), P(dres), acc, stream), bname + ".bwd")
        else:
            # (uses the shared reduction workspace `ws`: never from the branch stream, the main chain may be inside it)
            assert not tr._in_branch, "a branch-stream BatchNorm backward must come with its sums (Act.presums): it has no workspace of its own"
            _lib.check(lib.sp_bn_train_bwd_nhwc(P(ya.grad), gf, rs, P(z), P(mean), P(invstd), P(gamma), rows, C, P(dz), P(dgamma), P(dbeta),
                                                P(dres), acc, P(ws), stream), bname + ".bwd")
        in_branch = tr._in_branch
        if in_branch and not tr._arena_on:
            ya.grad.record_stream(tr._branch_stream)     # (allocator-owned, from the main stream's pool, read by the branch stream)
        ya.grad = None
        if res is not None:
            res.contrib += 1
        if in_branch:
            # on the branch stream: the weight-gradient job (queued with an event of the MAIN stream) and the bucket bookkeeping
            # wait for the join
            xa.deferred.append(lambda: self.wgrad_async(layer, xa.data, dz))
            xa.deferred.append(lambda: tr._grads_ready(bname + ".weight", bname + ".bias", cname + ".weight"))
        else:
            self.wgrad_async(layer, xa.data, dz)
        if xa.needs_grad and layer.need_dgrad:
            if not in_branch:
                self.join_grad(xa)
            # the last consumer to contribute sees the complete dy of xa in its epilogue: BN backward sums for free.  (Block outputs:
            # the residual share lands first, conv1 of the next block - a full-cover 1x1 - accumulates last.)
            last = xa.contrib == xa.consumers - 1
            fuse = (tr.fuse_bn_bwd and xa.bn is not None and last and (xa.grad is None or layer.dgrad_full_cover)
                    and layer.groups == 1)              # (a grouped launch has no BSTATS epilogue: that BatchNorm reduces its own sums)
            if xa.lazy_g is not None:
                assert fuse and xa.grad is None, "a lazy residual share needs the BSTATS dgrad of conv1 as the last contributor"
                xa.grad = layer.dgrad(dz, B, None, bn_src=xa, acc_masked=xa.lazy_g)
                xa.lazy_g = None
            else:
                xa.grad = layer.dgrad(dz, B, xa.grad, bn_src=xa if fuse else None)
            xa.contrib += 1
        if not in_branch:
            tr._grads_ready(bname + ".weight", bname + ".bias", cname + ".weight")

    # ---- SELayer ---------------------------------------------------------------------------------------------------------------------
    def se_gate(self, ua, idn, sname: str):
        """SELayer + the block tail (nets/commons.py:4-18, pose_resnet_dconv.py:124-131): y = relu(u * sigmoid(fc2(relu(fc0(mean_hw u)))) + identity).
        The FC layers are 1x1 convs on the pooled [B,1,1,C] map (conv kernels forward, backward and for the weight gradients)."""
        from .train import Act
        tr, lib, stream, B, bf = self.tr, self.lib, self.stream, self.B, self.bf
        new, newf = self.new, self.newf
        fc0, fc2 = self.L[sname + ".fc.0"], self.L[sname + ".fc.2"]
        u, C, hw = ua.data, ua.c, ua.h * ua.w
        ua.consumers += 1
        idn.consumers += 1
        sq = new((B, 1, 1, C))
        _lib.check((lib.sp_global_avg_pool_nhwc_bf16 if tr.bf16 else lib.sp_global_avg_pool_nhwc)(P(u), P(sq), B, hw, C, stream), sname + ".pool")
        hid = fc0.forward(sq, B, shift=tr.sd[sname + ".fc.0.bias"], relu=True)
        gl = fc2.forward(hid, B, shift=tr.sd[sname + ".fc.2.bias"])
        y = new(u.shape)
        _lib.check((lib.sp_se_gate_add_relu_nhwc_bf16 if tr.bf16 else lib.sp_se_gate_add_relu_nhwc)(P(u), P(gl), P(idn.data), P(y), B, hw, C,
                                                                                                   stream), sname + ".gate")
        ya = Act(y, ua.h, ua.w, C)

        def bwd():
            stream = self.stream
            dy = ya.grad
            da = newf((B, C))
            _lib.check(lib.sp_se_gate_bwd_reduce(P(dy), bf, P(y), P(u), B, hw, C, P(da), stream), sname + ".bwd")
            dg = new((B, 1, 1, C))
            _lib.check(lib.sp_se_sigmoid_bwd(P(da), bf, P(gl), B, C, P(dg), P(tr.flat.view(sname + ".fc.2.bias", True)), stream), sname + ".bwd")
            self.wgrad_async(fc2, hid, dg)
            dh = fc2.dgrad(dg, B, None)
            dhm = new((B, 1, 1, fc0.O))
            _lib.check(lib.sp_relu_bwd_rows(P(dh), bf, P(hid), B, fc0.O, P(dhm), P(tr.flat.view(sname + ".fc.0.bias", True)), stream), sname + ".bwd")
            self.wgrad_async(fc0, sq, dhm)
            ds = fc0.dgrad(dhm, B, None)
            acc = 0
            if idn.grad is None:
                idn.grad = newf(idn.data.shape)
            else:
                acc = 1
            ua.grad = newf(u.shape)
            _lib.check(lib.sp_se_gate_bwd_apply(P(dy), bf, P(y), P(gl), P(ds), B, hw, C, P(ua.grad), P(idn.grad), acc, stream), sname + ".bwd")
            ua.contrib += 1
            idn.contrib += 1
            ya.grad = None
            tr._grads_ready(sname + ".fc.0.weight", sname + ".fc.0.bias", sname + ".fc.2.weight", sname + ".fc.2.bias")
        self.tape.append(bwd)
        return ya

    # ---- HRNet fuse layers -----------------------------------------------------------------------------------------------------------
    def upsample_add(self, xa, base, f: int, relu: bool):
        """y = [relu](base + nearest_upsample(x, f)) (HRNet fuse layers, pose_hrnet.py:192-202,250-257; f = 1: the identity term)."""
        from .train import Act
        tr, lib, stream, B, bf = self.tr, self.lib, self.stream, self.B, self.bf
        xa.consumers += 1
        base.consumers += 1
        y = self.new(base.data.shape)
        _lib.check((lib.sp_upsample_add_nhwc_bf16 if tr.bf16 else lib.sp_upsample_add_nhwc)(P(xa.data), P(base.data), P(y), B, xa.h, xa.w, xa.c, f,
                                                                                           int(relu), stream), "fuse")
        ya = Act(y, base.h, base.w, base.c)

        def bwd():
            accs = []
            for t in (base, xa):
                accs.append(0 if t.grad is None else 1)
                if t.grad is None:
                    t.grad = self.newf(t.data.shape)
            _lib.check(lib.sp_upsample_add_bwd_nhwc(P(ya.grad), bf, P(y) if relu else None, B, xa.h, xa.w, xa.c, f, P(base.grad), accs[0],
                                                    P(xa.grad), accs[1], self.stream), "fuse.bwd")
            base.contrib += 1
            xa.contrib += 1
            ya.grad = None
        self.tape.append(bwd)
        return ya

    # ---- DUC head --------------------------------------------------------------------------------------------------------------------
    def shuffle(self, xa):
        """nn.PixelShuffle(2) and, on the tape, its inverse permutation for the gradient."""
        from .train import Act
        tr, lib, B = self.tr, self.lib, self.B
        xa.consumers += 1
        y = self.new((B, 2 * xa.h, 2 * xa.w, xa.c // 4))
        _lib.check((lib.sp_pixel_shuffle2_nhwc_bf16 if tr.bf16 else lib.sp_pixel_shuffle2_nhwc)(P(xa.data), P(y), B, xa.h, xa.w, xa.c,
                                                                                               self.stream), "pixel_shuffle")
        ya = Act(y, 2 * xa.h, 2 * xa.w, xa.c // 4)

        def bwd():
            assert xa.grad is None
            xa.grad = self.newg(xa.data.shape)
            _lib.check((lib.sp_pixel_unshuffle2_nhwc_bf16 if tr.g16 else lib.sp_pixel_unshuffle2_nhwc)(P(ya.grad), P(xa.grad), B, xa.h, xa.w, xa.c,
                                                                                                       self.stream), "pixel_shuffle.bwd")
            xa.contrib += 1
            ya.grad = None
        self.tape.append(bwd)
        return ya

    # ---- the ResNet stem -------------------------------------------------------------------------------------------------------------
    def input_nhwc(self, x: torch.Tensor):
        """[B,3,H,W] fp32 -> the stem's loader format (NHWC4 fp32 / NHWC8 bf16); the network input takes no gradient."""
        from .train import Act
        tr, lib, B = self.tr, self.lib, self.B
        cp = 8 if tr.bf16 else 4
        x4 = self.new((B, tr.in_h, tr.in_w, cp))
        _lib.check((lib.sp_nchw_to_nhwc8_bf16 if tr.bf16 else lib.sp_nchw_to_nhwc4)(P(x), P(x4), B, 3, tr.in_h, tr.in_w, self.stream), "to_nhwc")
        return Act(x4, tr.in_h, tr.in_w, cp, needs_grad=False)

    def stem_fused(self, xin):
        """conv1 -> bn1 -> relu -> maxpool with the BatchNorm map applied inside the pooling pass (sp_bn_apply_maxpool_nhwc): relu(bn1(z)) feeds the
        pooling only, so the 128 x 96 map is never written; backward sums d gamma / d beta over the POOLED grid and gathers dz
        (sp_bn_maxpool_bwd_nhwc) - no pooling input gradient either.  Same values as conv_bn + sp_maxpool3x3s2_idx_nhwc."""
        from .train import Act
        tr, lib, stream, B, bf, gf, ws, dev = self.tr, self.lib, self.stream, self.B, self.bf, self.gf, self.ws, self.dev
        layer = self.L["conv1"]
        xin.consumers += 1
        z, part, prow = layer.forward_bn_stats(xin.data, B)
        hs, wsz, C = z.shape[1], z.shape[2], z.shape[3]
        mean, invstd = self.newf(C), self.newf(C)
        gamma, beta = tr.sd["bn1.weight"], tr.sd["bn1.bias"]
        _lib.check(lib.sp_bn_train_stats_from_conv(P(part[0]), P(part[1]), prow, part.shape[2], B * hs * wsz, C, BN_EPS, BN_MOMENTUM, P(mean), P(invstd),
                                                   P(tr.buffers["bn1.running_mean"]), P(tr.buffers["bn1.running_var"]), stream), "bn1")
        self.nbt.append(tr.buffers["bn1.num_batches_tracked"])
        pooled = self.new((B, hs // 2, wsz // 2, C))
        pidx = tr._take(tuple(pooled.shape), torch.uint8, dev)
        _lib.check(lib.sp_bn_apply_maxpool_nhwc(P(z), bf, P(mean), P(invstd), P(gamma), P(beta), P(pooled), P(pidx), B, hs, wsz, C, stream), "bn1+maxpool")
        out = Act(pooled, hs // 2, wsz // 2, C)

        def bwd():
            dz = self.new(z.shape)
            _lib.check(lib.sp_bn_maxpool_bwd_nhwc(P(out.grad), gf, P(pidx), P(z), P(mean), P(invstd), P(gamma), P(beta), B, hs, wsz, C,
                                                  P(tr.flat.view("bn1.weight", True)), P(tr.flat.view("bn1.bias", True)), P(dz), P(ws), self.stream),
                       "bn1+maxpool.bwd")
            out.grad = None
            self.wgrad_async(layer, xin.data, dz)
            tr._grads_ready("bn1.weight", "bn1.bias", "conv1.weight")
        self.tape.append(bwd)
        return out

    def stem_plain(self, xin):
        """conv1 -> bn1 -> relu as a conv_bn, then the 3x3 stride-2 max pool with its winning taps kept for the backward gather."""
        from .train import Act
        tr, lib, B, bf, gf, dev = self.tr, self.lib, self.B, self.bf, self.gf, self.dev
        stem_out = self.conv_bn(xin, "conv1", "bn1", True)
        pooled = self.new((B, stem_out.h // 2, stem_out.w // 2, stem_out.c))
        pool_idx = tr._take(tuple(pooled.shape), torch.uint8, dev)               # winning tap per output element
        _lib.check(lib.sp_maxpool3x3s2_idx_nhwc(P(stem_out.data), bf, P(pooled), P(pool_idx), B, stem_out.h, stem_out.w, stem_out.c, self.stream), "maxpool")
        pa = Act(pooled, stem_out.h // 2, stem_out.w // 2, stem_out.c)

        def pool_bwd():
            stem_out.grad = self.newg(stem_out.data.shape)
            _lib.check(lib.sp_maxpool3x3s2_bwd_idx_nhwc(P(pool_idx), P(pa.grad), gf, P(stem_out.grad), B, stem_out.h, stem_out.w, stem_out.c,
                                                        self.stream), "maxpool.bwd")
            pa.grad = None
        self.tape.append(pool_bwd)
        return pa

    def stem(self, xin):
        tr = self.tr
        if tr.fuse_stem_pool and not self.sync and tr.fuse_bn_stats and tr.in_h % 4 == 0 and tr.in_w % 4 == 0:
            return self.stem_fused(xin)
        return self.stem_plain(xin)


# =============================================================================================================== per-family builders
def build_hrnet(t: StepTape, a):
    """PoseHighResolutionNet.forward after the first stem conv (pose_hrnet.py:419-454, :241-259, :181-236, :327-366)."""
    L = t.L
    extra = t.tr.model.cfg["MODEL"]["EXTRA"]
    a = t.conv_bn(a, "conv2", "bn2", True)
    for k in range(4):
        p = f"layer1.{k}"
        u = t.conv_bn(a, p + ".conv1", p + ".bn1", True)
        u = t.conv_bn(u, p + ".conv2", p + ".bn2", True)
        idn = t.conv_bn(a, p + ".downsample.0", p + ".downsample.1", False) if k == 0 else a
        a = t.conv_bn(u, p + ".conv3", p + ".bn3", True, res=idn)
    ys, pre_n = [a], 1
    for si, st in enumerate((2, 3, 4)):
        sc = extra[f"STAGE{st}"]
        nb = sc["NUM_BRANCHES"]
        tn = f"transition{si + 1}"
        xs: List = []
        for i in range(nb):
            if i < pre_n:
                xs.append(t.conv_bn(ys[i], f"{tn}.{i}.0", f"{tn}.{i}.1", True) if (f"{tn}.{i}.0") in L else ys[i])
            else:
                v = ys[-1]
                for j in range(i + 1 - pre_n):
                    v = t.conv_bn(v, f"{tn}.{i}.{j}.0", f"{tn}.{i}.{j}.1", True)
                xs.append(v)
        for m in range(sc["NUM_MODULES"]):
            multi = not (st == 4 and m == sc["NUM_MODULES"] - 1)
            base = f"stage{st}.{m}"
            for i in range(nb):
                for k in range(sc["NUM_BLOCKS"][i]):
                    p = f"{base}.branches.{i}.{k}"
                    u = t.conv_bn(xs[i], p + ".conv1", p + ".bn1", True)
                    xs[i] = t.conv_bn(u, p + ".conv2", p + ".bn2", True, res=xs[i])
            outs = []
            for i in range(nb if multi else 1):
                y = None
                for j in range(nb):
                    last = j == nb - 1
                    f = f"{base}.fuse_layers.{i}.{j}"
                    if j == i:
                        y = xs[i] if y is None else t.upsample_add(xs[i], y, 1, last)
                    elif j > i:
                        u = t.conv_bn(xs[j], f + ".0", f + ".1", False, shortcut=False)
                        y = t.upsample_add(u, y, 2 ** (j - i), last)
                    else:
                        u = xs[j]
                        for k in range(i - j):
                            fin = k == i - j - 1
                            u = t.conv_bn(u, f"{f}.{k}.0", f"{f}.{k}.1", last if fin else True, res=y if fin else None, shortcut=False)
                        y = u
                outs.append(y)
            xs = outs
        ys, pre_n = xs, nb
    return ys[0]


def build_resnet_bottleneck(t: StepTape, a):
    """layer1..4 of Bottlenecks (resnet50 / 101 / 152, wide_resnet*_2; pose_resnet_dconv.py:83-133, :205-221), SELayer variant included."""
    tr, L, sync, branch, dev = t.tr, t.L, t.sync, t.branch, t.dev
    for li, n in enumerate(tr.model.BLOCKS, start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            p1 = pdn = None
            if bi == 0 and sync:
                # conv1 and the projection shortcut read the same input: both convs first, ONE statistics all-reduce for the pair
                p1, pdn = t.conv_stats(a, p + ".conv1"), t.conv_stats(a, p + ".downsample.0")
                t.batch_stats([p1, pdn], [p + ".bn1", p + ".downsample.1"])
            join_fwd = None
            if bi == 0 and branch is not None:
                # projection shortcut on the branch stream; its tape entry keeps its old place (after conv2's, before conv3's)
                blk_in, i_ds = a, len(t.tape)
                idn, ev_ds = t.run_on_branch(lambda: t.conv_bn(blk_in, p + ".downsample.0", p + ".downsample.1", False))
                ds_bwd = t.tape.pop(i_ds)

                def ds_bwd_on_branch(ds_bwd=ds_bwd, blk_in=blk_in):
                    _, ev = t.run_on_branch(ds_bwd)
                    blk_in.grad_event = ev
                    tr._branch_open.append(blk_in)
                join_fwd = lambda ev_ds=ev_ds: torch.cuda.current_stream(dev).wait_event(ev_ds)
            if bi > 0 and len(L[p + ".conv1"].d_dgrad) == 1 and L[p + ".conv1"].dgrad_full_cover and (p + ".se.fc.0") not in L:
                a.lazy_ok = True                   # identity block: consumers = conv1 and the residual add
            u = t.conv_bn(a, p + ".conv1", p + ".bn1", True, pend=p1)
            u = t.conv_bn(u, p + ".conv2", p + ".bn2", True, defer_apply_to=p + ".conv3")     # bn2 + ReLU inside conv3's staging pass where it can be
            if join_fwd is not None:
                t.tape.append(ds_bwd_on_branch)
            else:
                idn = t.conv_bn(a, p + ".downsample.0", p + ".downsample.1", False, pend=pdn) if bi == 0 else a
            if (p + ".se.fc.0") in L:
                if idn.sibling is not None:
                    idn.sibling = None                 # the shortcut's consumer is the gate, not a BatchNorm epilogue: it reduces its own sums
                a = t.se_gate(t.conv_bn(u, p + ".conv3", p + ".bn3", False, shortcut=False), idn, p + ".se")
            else:
                a = t.conv_bn(u, p + ".conv3", p + ".bn3", True, res=idn, before_apply=join_fwd)
    return a


def build_resnet_basic(t: StepTape, a):
    """layer1..4 of BasicBlocks (resnet18 / resnet34; pose_resnet_dconv.py:38-80): conv3x3 (carries the stride) -> bn -> relu -> conv3x3 -> bn,
    + identity or the projection shortcut where the shape changes (not layer1.0), relu; SELayer on the blocks that have a shortcut."""
    tr, L, sync = t.tr, t.L, t.sync
    for li, n in enumerate(tr.model.BLOCKS, start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            has_ds = (p + ".downsample.0") in L
            p1 = pdn = None
            if has_ds and sync:
                p1, pdn = t.conv_stats(a, p + ".conv1"), t.conv_stats(a, p + ".downsample.0")
                t.batch_stats([p1, pdn], [p + ".bn1", p + ".downsample.1"])
            u = t.conv_bn(a, p + ".conv1", p + ".bn1", True, pend=p1)
            idn = t.conv_bn(a, p + ".downsample.0", p + ".downsample.1", False, pend=pdn) if has_ds else a
            if (p + ".se.fc.0") in L:
                if idn.sibling is not None:
                    idn.sibling = None
                a = t.se_gate(t.conv_bn(u, p + ".conv2", p + ".bn2", False, shortcut=False), idn, p + ".se")
            else:
                a = t.conv_bn(u, p + ".conv2", p + ".bn2", True, res=idn)
    return a


def build_head_dconv(t: StepTape, a):
    """Three ConvTranspose2d(4, 2, 1) + BatchNorm + ReLU (pose_resnet_dconv.py:236-249)."""
    for idx in (0, 3, 6):
        a = t.conv_bn(a, f"deconv_layers.{idx}", f"deconv_layers.{idx + 1}", True)
    return a


def build_head_duc(t: StepTape, a):
    """PixelShuffle -> DUC -> DUC (pose_resnet_duc.py:227-232, nets/commons.py:21-43: conv3x3 + BatchNorm + ReLU + PixelShuffle)."""
    a = t.shuffle(a)
    for idx in (1, 2):
        a = t.shuffle(t.conv_bn(a, f"duc_layers.{idx}.conv", f"duc_layers.{idx}.bn", True))
    return a


def record(tr, x: torch.Tensor):
    """Launch the train-mode forward of `tr.model` on x [B,3,H,W] up to (not including) final_layer; returns (last activation, tape)."""
    t = StepTape(tr, x)
    xin = t.input_nhwc(x)
    if tr.head == "hrnet":
        return build_hrnet(t, t.conv_bn(xin, "conv1", "bn1", True)), t
    a = t.stem(xin)
    a = (build_resnet_basic if getattr(tr.model, "BLOCK", "bottleneck") == "basic" else build_resnet_bottleneck)(t, a)
    a = (build_head_dconv if tr.head == "dconv" else build_head_duc)(t, a)
    return a, t
