"""TEST HELPER: CPU interpreter of the launch descriptors (include/simple_pose_hip.h semantics), in plain torch.

It executes a simple_pose_amd.engine.Program the way the HIP kernels are documented to, so that the HOST logic
(weight packing, descriptor geometry, buffer planning) can be verified against the oracle without a GPU.  It is not a
product path and not a fallback: simple_pose_amd never imports it.
"""
import torch

from simple_pose_amd._lib import SP_CONV_BF16, SP_CONV_OUT_NCHW, SP_CONV_PIXEL_SHUFFLE, SP_CONV_RELU


def conv_desc_cpu(d, x, w, scale, shift, res, y, B):
    """x: [B,in_h,in_w,c_in]; w: [phases*n_pad, k_pad]; y: NHWC [B,out_h,out_w,out_c] or NCHW if flagged (in place)."""
    phases = d.phases_y * d.phases_x
    x = x.reshape(B, d.in_h, d.in_w, d.c_in)      # the bf16 stem reads the [h, w, 4] image as pixel pairs [h, w/2, 8]
    wp = w.reshape(phases, d.n_pad, d.k_pad)
    gy = torch.arange(d.grid_h).view(-1, 1)
    gx = torch.arange(d.grid_w).view(1, -1)
    for ph in range(phases):
        py, px = ph // d.phases_x, ph % d.phases_x
        cols = []
        for ty in range(d.taps_h):
            for tx in range(d.taps_w):
                iy = gy * d.stride + d.dy0 + py + ty * d.dy_step
                ix = gx * (d.stride_x or d.stride) + d.dx0 + px + tx * d.dx_step
                ok = ((iy >= 0) & (iy < d.in_h) & (ix >= 0) & (ix < d.in_w))
                iyc, ixc = iy.clamp(0, d.in_h - 1).expand(d.grid_h, d.grid_w), ix.clamp(0, d.in_w - 1).expand(d.grid_h, d.grid_w)
                g = x[:, iyc, ixc, :].float() * ok.expand(d.grid_h, d.grid_w)[None, :, :, None]
                cols.append(g)
        if getattr(d, "c_in_group", 0):
            # grouped launch (sp_conv_desc.c_in_group): N tile t of c_in_group columns reads channels [t * g, (t + 1) * g) of every tap, K = taps * g
            g = d.c_in_group
            acc = torch.zeros(A0 := cols[0].shape[:-1] + (d.n_pad,), dtype=torch.float64)
            for t in range(d.c_out // g):
                At = torch.cat([c[..., t * g:(t + 1) * g] for c in cols], dim=-1)
                acc[..., t * g:(t + 1) * g] = At.double() @ wp[ph, t * g:(t + 1) * g, :At.shape[-1]].double().t()
        else:
            A = torch.cat(cols, dim=-1)  # [B,gh,gw,taps*c_in]
            k = A.shape[-1]
            acc = A.double() @ wp[ph, :, :k].double().t()  # [B,gh,gw,n_pad]
        acc = acc[..., :d.c_out]
        if scale is not None:
            acc = acc * scale.double()
        if shift is not None:
            acc = acc + shift.double()
        oy = (torch.arange(d.grid_h) * d.oy_mul + d.oy_add + py)
        ox = (torch.arange(d.grid_w) * d.ox_mul + d.ox_add + px)
        if d.flags & SP_CONV_PIXEL_SHUFFLE:
            oc = d.out_c
            for sub in range(4):
                part = acc[..., sub * oc:(sub + 1) * oc]
                yy, xx = oy + (sub >> 1), ox + (sub & 1)
                if res is not None:
                    part = part + res[:, yy][:, :, xx].double()
                if d.flags & SP_CONV_RELU:
                    part = part.clamp(min=0)
                y[:, yy.view(-1, 1), xx.view(1, -1), :] = part.float().to(y.dtype)
            continue
        if res is not None:
            acc = acc + res[:, oy][:, :, ox].double()
        if d.flags & SP_CONV_RELU:
            acc = acc.clamp(min=0)
        if d.flags & SP_CONV_OUT_NCHW:
            y[:, :, oy.view(-1, 1), ox.view(1, -1)] = acc.permute(0, 3, 1, 2).float()
        else:
            y[:, oy.view(-1, 1), ox.view(1, -1), :] = acc.float().to(y.dtype)


def run_program_cpu(prog, x):
    B = x.shape[0]
    act_dt = torch.bfloat16 if prog.dtype == "bf16" else torch.float32
    bufs = {"input": x}
    # a fused stem (`stem7`) is defined as its three-launch lowering (kept in the op for uint8 input; the GPU tests hold the fused
    # kernel to that lowering bit for bit): interpret those descriptors
    ops = [u for op in prog.ops for u in (op.args[-1] if op.kind in ("stem7", "hstem", "htrans") else (op,))]
    for op in ops:
        if op.kind == "to_nhwc4":
            c, h, w = op.args
            cp = prog.shapes[op.dst][2]
            t = torch.zeros((B, h, w, cp))
            t[..., :c] = bufs[op.src].permute(0, 2, 3, 1)
            bufs[op.dst] = t.to(act_dt)
        elif op.kind == "maxpool":
            t = torch.nn.functional.max_pool2d(bufs[op.src].float().permute(0, 3, 1, 2), 3, 2, 1).to(act_dt)
            bufs[op.dst] = t.permute(0, 2, 3, 1).contiguous()
        elif op.kind == "pixel_shuffle":
            t = torch.nn.functional.pixel_shuffle(bufs[op.src].float().permute(0, 3, 1, 2), 2).to(act_dt)
            bufs[op.dst] = t.permute(0, 2, 3, 1).contiguous()
        elif op.kind == "gap":
            bufs[op.dst] = bufs[op.src].mean(dim=(1, 2), keepdim=True)
        elif op.kind == "se_gate":
            hw, c, gate = op.args
            bufs[op.dst] = torch.relu(bufs[op.src] * torch.sigmoid(bufs[gate]) + bufs[op.res])
        elif op.kind == "upsample_add":
            h, w, c, f, relu = op.args
            up = bufs[op.src].float().repeat_interleave(f, 1).repeat_interleave(f, 2)
            t = bufs[op.res].float() + up
            bufs[op.dst] = (t.clamp(min=0) if relu else t).to(act_dt)
        elif op.kind == "upsample_add_n":
            H, W, c, relu, more, factors = op.args
            t = bufs[op.res].float()
            for name, f in zip((op.src,) + tuple(more), factors):                # the reference's order of additions, fp32, one rounding
                t = t + bufs[name].float().repeat_interleave(f, 1).repeat_interleave(f, 2)
            bufs[op.dst] = (t.clamp(min=0) if relu else t).to(act_dt)
        elif op.kind == "conv":
            d = op.desc
            d.batch = B
            if op.dst == prog.out_name:
                y = torch.full((B,) + tuple(prog.out_shape), float("nan"))
            else:
                y = torch.full((B, d.out_h, d.out_w, d.out_c), float("nan"), dtype=act_dt)
            conv_desc_cpu(d, bufs[op.src], op.w, op.scale, op.shift, bufs[op.res] if op.res else None, y, B)
            assert not torch.isnan(y).any(), f"{op.name}: launch does not cover its output"
            bufs[op.dst] = y
        elif op.kind == "dual1x1":
            # y = relu(bn3(t . W3^T) + bn_d(x . Wd^T)) (sp_dual_pw_bf16 / sp_dual_pw_f32): the shortcut value takes the activation dtype's rounding,
            # as the tensor the two-launch program stores
            w_s, s_s, h_s, rows_per_image, relu = op.args
            t, xs = bufs[op.src].double(), bufs[op.res].double()
            w3, wd = op.w.double()[:256, :64], w_s.double()[:256, :64]
            one = lambda v, n: torch.ones(n, dtype=torch.float64) if v is None else v.double()
            zero = lambda v, n: torch.zeros(n, dtype=torch.float64) if v is None else v.double()
            r = (xs @ wd.T) * one(s_s, 256) + zero(h_s, 256)
            r = r.float().to(act_dt).double()
            y = (t @ w3.T) * one(op.scale, 256) + zero(op.shift, 256) + r
            if relu:
                y = y.clamp(min=0)
            bufs[op.dst] = y.float().to(act_dt)
        else:
            raise ValueError(op.kind)
    return bufs[prog.out_name], bufs


class TorchPacker:
    """TEST HELPER: the packed layouts of include/simple_pose_hip.h (sp_pack_conv_weights, sp_pack_deconv_k4s2p1, sp_fold_bn) restated
    in plain torch, so that the CPU-only host-logic tests can lower a network without a GPU, and so that the GPU tests have an
    independent statement of the layouts to hold the device kernels against.  Same interface as simple_pose_amd.engine.HipPacker."""

    @staticmethod
    def _round_up(v, m):
        return (v + m - 1) // m * m

    @classmethod
    def n_pad_for(cls, c_out):
        return cls._round_up(c_out, 128) if c_out >= 128 else (cls._round_up(c_out, 64) if c_out > 32 else 32)

    def conv(self, w, *, c_in_pad=None, taps_w_pad=None, pixel_shuffle=False, pair_s0=-1, bf16=False):
        w = w.detach().float()
        O, I, kh, kw = w.shape
        if pair_s0 >= 0:                      # x-paired stem: channel sub*4 + c of pair pt holds pixel tap kx = 2*pt + sub - s0
            w2 = torch.zeros((O, 8, kh, taps_w_pad), dtype=torch.float32, device=w.device)
            for kx in range(kw):
                pt, sub = (kx + pair_s0) // 2, (kx + pair_s0) % 2
                w2[:, sub * 4: sub * 4 + I, :, pt] = w[:, :, :, kx]
            w, I, kw = w2, 8, taps_w_pad
        ci, tw = c_in_pad or I, taps_w_pad or kw
        p = torch.zeros((O, kh, tw, ci), dtype=torch.float32, device=w.device)
        p[:, :, :kw, :I] = w.permute(0, 2, 3, 1)
        p = p.reshape(O, kh * tw * ci)
        if pixel_shuffle:
            p = p[self.row_perm(O, w.device)]
        k = kh * tw * ci
        k_pad, n_pad = self._round_up(k, 64 if bf16 else 32), self.n_pad_for(O)
        out = torch.zeros((n_pad, k_pad), dtype=torch.float32, device=w.device)
        out[:O, :k] = p
        return (out.to(torch.bfloat16) if bf16 else out).contiguous(), kh, tw, ci, k_pad

    def grouped(self, w, groups, panel, *, bf16=False):
        """sp_pack_conv_weights_grouped restated: [O, O/g, kh, kw] -> [O, kh*kw*panel], row n's panel = channels [(n // panel) * panel, +panel)."""
        wf = w.detach().float()
        O, cpg, kh, kw = wf.shape
        out = torch.zeros((O, kh * kw, panel), dtype=torch.float32, device=w.device)
        for n in range(O):
            lo = (n // cpg) * cpg - (n // panel) * panel          # the group's first channel inside the panel
            out[n, :, lo:lo + cpg] = wf[n].reshape(cpg, kh * kw).t()
        out = out.reshape(O, kh * kw * panel)
        return (out.to(torch.bfloat16) if bf16 else out).contiguous()

    def deconv(self, w, *, bf16=False):
        wf = w.detach().float()
        I, O = wf.shape[:2]
        n_pad = self.n_pad_for(O)
        out = torch.zeros((4, n_pad, 4 * I), dtype=torch.float32, device=w.device)
        for py in range(2):
            for px in range(2):
                for ty in range(2):
                    for tx in range(2):
                        ky, kx = 2 * ty + 1 - py, 2 * tx + 1 - px
                        t = ty * 2 + tx
                        out[py * 2 + px, :O, t * I:(t + 1) * I] = wf[:, :, ky, kx].t()
        out = out.reshape(4 * n_pad, 4 * I)
        return (out.to(torch.bfloat16) if bf16 else out).contiguous(), n_pad

    @staticmethod
    def row_perm(c_out, device):
        """packed row n' = sub*(C/4) + c holds original channel c*4 + sub (sub = i*2 + j of nn.PixelShuffle(2))"""
        c4 = c_out // 4
        n = torch.arange(c_out, device=device)
        return (n % c4) * 4 + (n // c4)

    def fold_bn(self, weight, bias, running_mean, running_var, eps=1e-5, pixel_shuffle=False):
        """scale = w / sqrt(var + eps), shift = b - mean * scale, every operation an IEEE fp32 operation of its own (numpy: the
        vectorised torch-CPU expression 1.0 / torch.sqrt(x) is NOT correctly rounded on every host - it differed by 1 ulp on the GPU
        box's AVX-512 cores - so it cannot be the bit-exact checker of sp_fold_bn)."""
        import numpy as np
        f = lambda t: t.detach().float().cpu().numpy()
        w, b, m, v = f(weight), f(bias), f(running_mean), f(running_var)
        root = np.sqrt((v + np.float32(eps)).astype(np.float32).astype(np.float64)).astype(np.float32)   # correctly rounded fp32 root
        scale = (w * (np.float32(1.0) / root).astype(np.float32)).astype(np.float32)
        shift = (b - (m * scale).astype(np.float32)).astype(np.float32)
        scale, shift = torch.from_numpy(scale), torch.from_numpy(shift)
        if pixel_shuffle:
            perm = self.row_perm(scale.numel(), "cpu")
            scale, shift = scale[perm], shift[perm]
        return scale.contiguous().to(weight.device), shift.contiguous().to(weight.device)

    def bias(self, b):
        return b.detach().float().contiguous()
