"""Experiment: what would an intra-workgroup split-K buy on the few-tile layers of the 32-image train step?  Times the STATS forward launch of a
layer with its full reduction and with half of it (half the input channels) on every tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_backward_kernels import _OneLayer, DEV
from simple_pose_amd import _lib

B = 32
cases = [("layer4.conv2", "conv", 512, 512, 3, 1, 1, 8, 6), ("layer3.conv2", "conv", 256, 256, 3, 1, 1, 16, 12),
         ("layer4.conv1", "conv", 2048, 512, 1, 1, 0, 8, 6), ("layer4.conv3", "conv", 512, 2048, 1, 1, 0, 8, 6),
         ("layer3.conv1", "conv", 1024, 256, 1, 1, 0, 16, 12), ("layer2.conv2", "conv", 128, 128, 3, 1, 1, 32, 24),
         ("deconv0", "deconv", 2048, 256, 4, 2, 1, 8, 6)]
for name, kind, I, O, k, s, p, H, W in cases:
    for frac in (1, 2):
        Ii = I // frac
        w = torch.randn((O, Ii, k, k) if kind == "conv" else (Ii, O, k, k)) * 0.05
        one = _OneLayer(kind, w, H, W, True, stride=s, pad=p)
        L = one.layer
        xs = [torch.randn(B, H, W, Ii, device=DEV).to(torch.bfloat16) for _ in range(4)]
        best = None
        for tm, tn in _lib.CONV_TILES:
            if L.d_fwd.n_pad % tn:
                continue
            L.d_fwd.tile_m, L.d_fwd.tile_n = tm, tn
            try:
                L.forward_bn_stats(xs[0], B)
            except Exception:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                L.forward_bn_stats(xs[i % 4], B)
            e1.record(); e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            if best is None or us < best[0]:
                best = (us, tm, tn)
        gf = 2 * B * L.oh * L.ow * O * Ii * (k * k if kind == "conv" else 4) / 1e9 if kind == "conv" else 2 * B * H * W * Ii * O * 16 / 1e9
        print(f"{name:14s} K/{frac}: best {best[0]:6.1f} us tile {best[1]}x{best[2]}  {gf:6.2f} GF {gf / best[0] * 1e-3 * 1e3:6.1f} TF/s", flush=True)
