#!/usr/bin/env python3
"""Diagnostic (not shipped): where the waves of bottleneck_c64_kernel spend their cycles, from the SP_BNECK_DIAG build of conv_bneck.hip.

    python tools/diag_bneck.py --build        # here: simple_pose_amd/lib/libsimple_pose_hip_bneckdiag.so (travels with gpurun)
    python tools/diag_bneck.py                # on the GPU box: per-role cycle breakdown of one launch at bs=128, 64x48, and its time
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "simple_pose_amd", "lib", "libsimple_pose_hip_bneckdiag.so")


def build():
    from simple_pose_amd import build as b
    b.build()
    obj = "/tmp/conv_bneck_diag.o"
    subprocess.run([b.HIPCC, "-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-c", "-DSP_BNECK_DIAG", "-I" + os.path.join(ROOT, "include"),
                    "-I" + b.CSRC, os.path.join(b.CSRC, "conv_bneck.hip"), "-o", obj], check=True)
    objs = [o for o in glob.glob(os.path.join(b.LIB_DIR, "*.o")) if os.path.basename(o) not in ("conv_bneck.o", "conv_ring_diag.o")]
    subprocess.run([b.HIPCC, f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [obj], check=True)
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    if a.build:
        return build()
    # SIMPLE_POSE_HIP_LIB already set (another build of the library, e.g. a same-box A/B): time that one, no stamps
    diag = os.path.isfile(LIB) and os.environ.get("SIMPLE_POSE_HIP_LIB", LIB) == LIB
    if diag:
        os.environ["SIMPLE_POSE_HIP_LIB"] = LIB
    import numpy as np
    import torch
    from simple_pose_amd import _lib, engine

    lib, dev, P = _lib.lib(), "cuda:0", _lib.ptr
    B, H, W = a.batch, 64, 48
    b = engine.ProgramBuilder(H, W, dtype="bf16")
    b.fuse_bottlenecks = True
    b.p.shapes["input"] = (H, W, 256)
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(dev)
    sc = lambda c: ((torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.1).to(dev))
    (s1, h1), (s2, h2), (s3, h3) = sc(64), sc(64), sc(256)
    out = b.bottleneck_c64("input", mk(64, 256, 1, 1), s1, h1, mk(64, 64, 3, 3), s2, h2, mk(256, 64, 1, 1), s3, h3, name="blk")
    op = b.p.ops[-1]
    x = torch.randn(B, H, W, 256, device=dev).bfloat16()
    y = torch.empty_like(x)
    bufs = {"input": x, out: y}
    st = _lib.current_stream()
    for _ in range(3):
        b.p._launch(lib, op, bufs, B, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.p._launch(lib, op, bufs, B, st)
    e1.record(); e1.synchronize()
    print(f"bottleneck_c64 bs={B}: {1e3 * e0.elapsed_time(e1) / 10:.1f} us per launch ({'diag build' if diag else os.path.basename(os.environ.get('SIMPLE_POSE_HIP_LIB', 'shipped build'))})")
    if diag:
        fn = ctypes.CDLL(LIB).sp_bneck_debug_read
        n = 256 * 8 * 8
        buf = (ctypes.c_ulonglong * n)()
        fn(buf, n)
        d = np.array(buf[:], dtype=np.float64).reshape(256, 8, 8)
        w = d[:, :4, :]
        tiles = np.median(w[:, :, 7])
        labels = ["stage A loop", "t1 store", "barrier waits", "stage B loop", "stage C passes", "next tile x requests"]
        print(f"tiles per workgroup {tiles:.0f}; cycles per tile (median over workgroups and waves): lifetime {np.median(w[:, :, 6]) / tiles:.0f}  " +
              "  ".join(f"{l} {np.median(w[:, :, i]) / tiles:.0f}" for i, l in enumerate(labels)))


if __name__ == "__main__":
    main()
