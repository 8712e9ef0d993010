#!/usr/bin/env python3
"""Diagnostic (not shipped): where a ring-kernel wave spends its cycles, from the SP_RING_DIAG build of conv_ring.hip.

    python tools/diag_ring.py --build                       # here: simple_pose_amd/lib/libsimple_pose_hip_ringdiag.so (travels with gpurun)
    python tools/diag_ring.py                                # on the GPU box: prints the per-stage breakdown of a few layers / tiles

Per wave (s_memtime stamps, one lane): cycles at the counted vmcnt wait, at the stage barrier, in the fragment-read + MFMA section, in
the epilogues; stages and tiles; in-kernel clock."""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "simple_pose_amd", "lib", "libsimple_pose_hip_ringdiag.so")


def build():
    from simple_pose_amd import build as b
    b.build()
    obj = os.path.join(b.LIB_DIR, "conv_ring_diag.o")
    subprocess.run([b.HIPCC, "-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-c", "-DSP_RING_DIAG", "-I" + os.path.join(ROOT, "include"),
                    "-I" + b.CSRC, os.path.join(b.CSRC, "conv_ring.hip"), "-o", obj], check=True)
    objs = [o for o in glob.glob(os.path.join(b.LIB_DIR, "*.o")) if os.path.basename(o) not in ("conv_ring.o", "conv_ring_diag.o")]
    subprocess.run([b.HIPCC, f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [obj], check=True)
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    a = ap.parse_args()
    if a.build:
        return build()
    os.environ["SIMPLE_POSE_HIP_LIB"] = LIB
    import numpy as np
    import torch
    from simple_pose_amd import _lib, engine

    lib = _lib.lib()
    dev = "cuda:0"

    def run(name, B, cin, h, w, cout, k, s, p, tiles, deconv=False, res=False):
        b = engine.ProgramBuilder(h, w, dtype="bf16")
        b.p.shapes["input"] = (h, w, cin)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        if deconv:
            out = b.deconv_k4s2p1("input", torch.randn(cin, cout, 4, 4, device=dev) * 0.02, scale=sc, shift=sh, relu=True)
        else:
            if res:
                b.p.shapes["res"] = ((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1, cout)
            out = b.conv("input", torch.randn(cout, cin, k, k, device=dev) * 0.02, stride=s, pad=p, scale=sc, shift=sh, relu=True, res="res" if res else None)
        op = b.p.ops[-1]
        d = op.desc
        d.batch = B
        x = torch.randn(B, h, w, cin, device=dev).bfloat16()
        y = torch.empty((B,) + tuple(b.p.shapes[out]), dtype=torch.bfloat16, device=dev)
        r = torch.randn_like(y) if res else None
        P = _lib.ptr
        for bm, bn in tiles:
            d.tile_m, d.tile_n, d.kernel = bm, bn, _lib.SP_CONV_KERNEL_RING
            if not lib.sp_conv2d_ring_ok(d):
                continue
            args = (d, P(x), P(op.w), P(op.scale), P(op.shift), P(r), P(y), _lib.current_stream())
            for _ in range(5):
                _lib.check(lib.sp_conv2d_fwd(*args))
            torch.cuda.synchronize()
            lib.sp_ring_debug_clear()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.sp_conv2d_fwd(*args); e1.record(); torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
            lib.sp_ring_debug_read(buf, 256 * 8 * 8)
            v = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
            v = v[v[:, 0, 4] > 0]
            st = v[:, :, 4]
            per = [(v[:, :, i] / st).mean() for i in range(4)]
            life, ticks, tl = v[:, :, 5].mean(), v[:, :, 6].mean(), v[:, :, 7].mean()
            flops = 2.0 * B * (h * w if deconv else ((h + 2 * p - k) // s + 1) * ((w + 2 * p - k) // s + 1)) * cin * cout * (16 if deconv else k * k)
            us = e0.elapsed_time(e1) * 1e3
            nm = 4 * (bm // (32 * (2 if (bm, bn) not in ((256, 128), (256, 64)) else 4))) * (bn // (32 * (4 if (bm, bn) not in ((256, 128), (256, 64)) else 2)))
            print(f"{name} tile {bm}x{bn}: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s; workgroups {len(v)}, stages/wave {st.mean():.0f}, tiles/wave {tl:.1f}; "
                  f"clock {life / ticks * 0.1:.2f} GHz")
            print(f"    cycles per stage per wave: vmcnt wait {per[0]:.0f} | barrier {per[1]:.0f} | frag reads + {nm} MFMAs (= {nm * 32} cyc of pipe) {per[2]:.0f} | "
                  f"epilogue share {per[3]:.0f}  -> {sum(per):.0f} total; lifetime {life:.0f} cycles")

    big = ((256, 256), (128, 256), (192, 128), (128, 128))
    run("deconv_layers.6 (256->256, 32x24 in)", 128, 256, 32, 24, 256, 4, 2, 1, big, deconv=True)
    run("duc_layers.2 (3x3 256->512 at 32x24)", 128, 256, 32, 24, 512, 3, 1, 1, big)
    run("layer3 conv2 (3x3 256->256 at 16x12)", 128, 256, 16, 12, 256, 3, 1, 1, big)
    run("layer3 conv3 (1x1 256->1024 + res at 16x12)", 128, 256, 16, 12, 1024, 1, 1, 0, big, res=True)
    run("layer4 conv2 (3x3 512->512 at 8x6)", 128, 512, 8, 6, 512, 3, 1, 1, big)


if __name__ == "__main__":
    main()
