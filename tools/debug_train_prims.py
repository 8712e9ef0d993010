"""Debug: ConvT forward / dgrad / wgrad and BN kernels in isolation against torch CPU autograd."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from simple_pose_amd import _lib, synth
from simple_pose_amd.train import ConvT, PackJob, _i32, _i64, P
dev = "cuda:0"
lib = _lib.lib()

class FakeFlat:
    def __init__(self): self.t = {}
    def view(self, name, grad=False): return self.t[name + ("/g" if grad else "")]
class FakeTr: pass

def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().pow(2).mean().sqrt() + 1e-30))

def run(kind, B, cin, cout, h, w, k, s, p, bf16=False):
    tr = FakeTr(); tr.bf16 = bf16; tr.flat = FakeFlat(); tr.wgrad_ws = torch.empty(48 * 1024 * 1024, device=dev)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, 4, 4)
    W = torch.from_numpy(synth.tensor_normal(1, f"w{wshape}", wshape, std=0.05))
    Wd = W.to(dev).contiguous()
    tr.flat.t["L.weight"] = Wd.view(-1); tr.flat.t["L.weight/g"] = torch.full_like(Wd.view(-1), float("nan"))
    layer = ConvT(tr, "L", kind, Wd, h, w, stride=s, pad=p)
    for j in layer.pack_jobs:
        _lib.check(lib.sp_permute4_f32(P(Wd), P(j.dst), int(bf16), _i32(*j.dims), _i64(*j.strides), _i32(*j.valid), j.base, j.dst_off, _lib.current_stream()))
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf16 else (lambda t: t)
    adt = torch.bfloat16 if bf16 else torch.float32
    x = rnd(torch.from_numpy(synth.tensor_normal(2, "x", (B, cin, h, w)))).requires_grad_(True)
    Wc = rnd(W.clone()).requires_grad_(True)
    y = F.conv2d(x, Wc, stride=s, padding=p) if kind == "conv" else F.conv_transpose2d(x, Wc, stride=2, padding=1)
    gy = rnd(torch.from_numpy(synth.tensor_normal(3, "gy", tuple(y.shape))))
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).to(adt)
    z = layer.forward(xd, B).float()
    gz = gy.permute(0, 2, 3, 1).contiguous().to(dev).to(adt)
    dx = layer.dgrad(gz, B, None).float()
    base = rnd(torch.from_numpy(synth.tensor_normal(4, "base", tuple(dx.shape)))).to(dev)
    dx2 = layer.dgrad(gz, B, base.clone()).float()
    layer.wgrad(xd, gz, B)
    torch.cuda.synchronize()
    print(("bf16 " if bf16 else "fp32 ") + f"{kind} B{B} {cin}->{cout} {h}x{w} k{k}s{s}p{p}: fwd {rel(z.cpu().permute(0,3,1,2), y.detach()):.2e} dgrad {rel(dx.cpu().permute(0,3,1,2), x.grad):.2e} "
          f"dgrad+acc {rel((dx2-base).cpu().permute(0,3,1,2), x.grad):.2e} wgrad {rel(tr.flat.t['L.weight/g'].view(wshape).cpu(), Wc.grad):.2e}")

for args in [("conv", 3, 256, 1024, 6, 4, 1, 1, 0), ("conv", 3, 1024, 256, 6, 4, 1, 1, 0), ("conv", 3, 256, 256, 6, 4, 3, 1, 1),
             ("conv", 3, 256, 256, 12, 8, 3, 2, 1), ("conv", 3, 512, 1024, 12, 8, 1, 2, 0), ("conv", 2, 64, 64, 16, 16, 3, 1, 1),
             ("conv", 3, 1024, 2048, 6, 4, 1, 2, 0), ("conv", 3, 512, 512, 6, 4, 3, 2, 1), ("deconv", 3, 2048, 256, 3, 2, 4, 2, 1),
             ("deconv", 2, 256, 256, 8, 8, 4, 2, 1)]:
    run(*args)
    run(*args, bf16=True)

# BN kernels
for (rows, C) in [(72, 1024), (72, 256), (1000, 64), (18, 2048), (4096, 512)]:
    z = torch.from_numpy(synth.tensor_normal(5, f"z{rows}{C}", (rows, C))) * 2 + 0.5
    res = torch.from_numpy(synth.tensor_normal(6, "r", (rows, C)))
    gam = torch.from_numpy(synth.tensor_uniform(7, "g", (C,), 0.5, 1.5)); bet = torch.from_numpy(synth.tensor_normal(8, "b", (C,)))
    zt = z.clone().requires_grad_(True); gt = gam.clone().requires_grad_(True); bt = bet.clone().requires_grad_(True); rt = res.clone().requires_grad_(True)
    y = F.relu(F.batch_norm(zt.t().reshape(1, C, rows, 1), None, None, gt, bt, True, 0.1, 1e-5).reshape(C, rows).t() + rt)
    gy = torch.from_numpy(synth.tensor_normal(9, "gy", (rows, C)))
    y.backward(gy)
    ws = torch.empty(256 * 2048 * 2, dtype=torch.float64, device=dev)
    zd, rd, gd, bd, gyd = (t.to(dev).contiguous() for t in (z, res, gam, bet, gy))
    mean, invstd, yd = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(rows, C, device=dev)
    _lib.check(lib.sp_bn_train_stats_nhwc(P(zd), 0, rows, C, 1e-5, 0.1, P(mean), P(invstd), None, None, P(ws), _lib.current_stream()))
    _lib.check(lib.sp_bn_apply_nhwc(P(zd), 0, P(mean), P(invstd), P(gd), P(bd), P(rd), P(yd), rows, C, 1, _lib.current_stream()))
    dz, dg, db, dr = torch.empty(rows, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(rows, C, device=dev)
    _lib.check(lib.sp_bn_train_bwd_nhwc(P(gyd), 0, P(yd), P(zd), P(mean), P(invstd), P(gd), rows, C, P(dz), P(dg), P(db), P(dr), 0, P(ws), _lib.current_stream()))
    torch.cuda.synchronize()
    print(f"bn rows {rows} C {C}: y {rel(yd.cpu(), y.detach()):.2e} dz {rel(dz.cpu(), zt.grad):.2e} dgamma {rel(dg.cpu(), gt.grad):.2e} dbeta {rel(db.cpu(), bt.grad):.2e} dres {rel(dr.cpu(), rt.grad):.2e}")
