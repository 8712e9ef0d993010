#!/usr/bin/env python3
"""Diagnostic (not shipped): where the waves of stem_pool_kernel spend their cycles, from the SP_STEM_DIAG build of conv_stem.hip.

    python tools/diag_stem.py --build        # here: simple_pose_amd/lib/libsimple_pose_hip_stemdiag.so (travels with gpurun)
    python tools/diag_stem.py [--dtype bf16] # on the GPU box: per-phase cycle breakdown of one launch at bs=128, 256x192, and its time
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "simple_pose_amd", "lib", "libsimple_pose_hip_stemdiag.so")


def build(extra=()):
    from simple_pose_amd import build as b
    b.build()
    obj = "/tmp/conv_stem_diag.o"
    subprocess.run([b.HIPCC, "-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-c", "-DSP_STEM_DIAG", *extra, "-I" + os.path.join(ROOT, "include"),
                    "-I" + b.CSRC, os.path.join(b.CSRC, "conv_stem.hip"), "-o", obj], check=True)
    objs = [o for o in glob.glob(os.path.join(b.LIB_DIR, "*.o")) if os.path.basename(o) not in ("conv_stem.o", "conv_ring_diag.o")]
    subprocess.run([b.HIPCC, f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [obj], check=True)
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D macros of an experiment build (e.g. SP_STEM_NOSTORE)")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dtype", default="both", choices=["fp32", "bf16", "both"])
    a = ap.parse_args()
    if a.build:
        return build(tuple("-D" + d for d in a.define))
    diag = os.path.isfile(LIB)
    if diag:
        os.environ["SIMPLE_POSE_HIP_LIB"] = LIB
    import numpy as np
    import torch
    from simple_pose_amd import _lib, engine

    lib, dev = _lib.lib(), "cuda:0"
    B, H, W = a.batch, 256, 192
    g = torch.Generator().manual_seed(0)
    w = (torch.randn((64, 3, 7, 7), generator=g) * 0.1).to(dev)
    scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    x = torch.randn((B, 3, H, W), generator=g).to(dev)
    for dtype in (("fp32", "bf16") if a.dtype == "both" else (a.dtype,)):
        b = engine.ProgramBuilder(H, W, dtype)
        out = b.stem_pool("input", w, scale, shift)
        op = b.p.ops[-1]
        assert op.kind == "stem7"
        bufs = dict(b.p._alloc(B, x.device))
        bufs["input"] = x
        st = _lib.current_stream()
        for _ in range(3):
            b.p._launch(lib, op, bufs, B, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.p._launch(lib, op, bufs, B, st)
        e1.record(); e1.synchronize()
        print(f"stem7 {dtype} bs={B}: {1e3 * e0.elapsed_time(e1) / 10:.1f} us per launch ({'diag build' if diag else 'shipped build'})")
        if diag:
            fn = ctypes.CDLL(LIB).sp_stem_debug_read
            n = 512 * 4 * 8
            buf = (ctypes.c_ulonglong * n)()
            fn(buf, n)
            d = np.array(buf[:], dtype=np.float64).reshape(512, 4, 8)
            d = d[d[:, 0, 7] > 0]
            tiles = np.median(d[:, :, 7])
            labels = ["park + barrier", "next patch requests", "GEMM rows + BN + conv tile store", "barrier", "pooling"]
            print(f"  {len(d)} workgroups, tiles per workgroup {tiles:.0f}; cycles per tile (median over workgroups, per wave):")
            for wv in range(4):
                print(f"    wave {wv}: lifetime {np.median(d[:, wv, 6]) / tiles:.0f}  " +
                      "  ".join(f"{l} {np.median(d[:, wv, i]) / tiles:.0f}" for i, l in enumerate(labels)))


if __name__ == "__main__":
    main()
