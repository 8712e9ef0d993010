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
    for op in prog.ops:
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
        else:
            raise ValueError(op.kind)
    return bufs[prog.out_name], bufs
