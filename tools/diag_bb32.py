#!/usr/bin/env python3
"""Diagnostic (not shipped): one HRNet BasicBlock of the 32-channel branch at bs=128, 64x48 - the fused launch (sp_basic_block_c32; SP_BB32_W8=0
selects the round-2 four-wave kernel) against the two conv launches it replaces, by HIP events on one stream.

    python tools/diag_bb32.py [--batch 128] [--h 64] [--w 48]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--w", type=int, default=48)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--channels", type=int, default=32, help="32 (sp_basic_block_c32) or 64 (sp_basic_block_c64; HRNet's map is --h 32 --w 24)")
    ap.add_argument("--build", action="store_true", help="here: the SP_BB32_DIAG build of the library (simple_pose_amd/lib/libsimple_pose_hip_bb32diag.so)")
    ap.add_argument("--variant", default="", help="suffix of the diag library's name (with --defs: another build of the kernel, for same-box A/Bs)")
    ap.add_argument("--defs", default="", help="extra compiler flags of the diag build, e.g. '-DBB_STAGGER=7'")
    ap.add_argument("--stamps", action="store_true", help="on the GPU box: per-phase cycle sums from the diag build")
    a = ap.parse_args()
    LIB = os.path.join(ROOT, "simple_pose_amd", "lib", f"libsimple_pose_hip_bb32diag{a.variant}.so")
    if a.build:
        import glob
        import subprocess
        from simple_pose_amd import build as b
        b.build()
        obj = "/tmp/conv_block_diag.o"
        subprocess.run([b.HIPCC, "-O3", f"--offload-arch={b.ARCH}", "-std=c++17", "-fPIC", "-c", "-DSP_BB32_DIAG"] + a.defs.split() + ["-I" + os.path.join(ROOT, "include"),
                        "-I" + b.CSRC, os.path.join(b.CSRC, "conv_block.hip"), "-o", obj], check=True)
        objs = [o for o in glob.glob(os.path.join(b.LIB_DIR, "*.o")) if os.path.basename(o) not in ("conv_block.o", "conv_ring_diag.o")]
        subprocess.run([b.HIPCC, f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + [obj], check=True)
        print(LIB)
        return
    if a.stamps:
        os.environ["SIMPLE_POSE_HIP_LIB"] = LIB
    import torch
    from simple_pose_amd import _lib, engine

    lib, dev = _lib.lib(), "cuda:0"
    B, H, W = a.batch, a.h, a.w
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.06).to(dev)
    sc = lambda c: ((torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.1).to(dev))
    CH = a.channels
    w1, w2, (s1, h1), (s2, h2) = mk(CH, CH, 3, 3), mk(CH, CH, 3, 3), sc(CH), sc(CH)
    x = torch.randn(B, H, W, CH, device=dev).bfloat16()
    res = {}
    outs = {}
    for fuse in (True, False):
        b = engine.ProgramBuilder(H, W, dtype="bf16")
        b.fuse_blocks = b.fuse_blocks64 = fuse
        b.p.shapes["input"] = (H, W, CH)
        y = b.basic_block_c32("input", w1, s1, h1, w2, s2, h2, name="blk")
        if not fuse:
            t = b.conv("input", w1, pad=1, scale=s1, shift=h1, relu=True, name="c1")
            y = b.conv(t, w2, pad=1, scale=s2, shift=h2, relu=True, res="input", name="c2")
        bufs = dict(b.p._alloc(B, torch.device(dev)))
        bufs["input"] = x
        st = _lib.current_stream()
        for _ in range(3):
            for op in b.p.ops:
                b.p._launch(lib, op, bufs, B, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            for op in b.p.ops:
                b.p._launch(lib, op, bufs, B, st)
        e1.record(); e1.synchronize()
        res[fuse] = 1e3 * e0.elapsed_time(e1) / a.reps
        outs[fuse] = bufs[y].clone()
    if a.stamps:
        import ctypes
        import numpy as np
        b = engine.ProgramBuilder(H, W, dtype="bf16")
        b.fuse_blocks = True
        b.p.shapes["input"] = (H, W, CH)
        y = b.basic_block_c32("input", w1, s1, h1, w2, s2, h2, name="blk")
        bufs = dict(b.p._alloc(B, torch.device(dev)))
        bufs["input"] = x
        for _ in range(4):                      # (the stamps are those of the last launch: warm caches)
            b.p._launch(lib, b.p.ops[0], bufs, B, _lib.current_stream())
        torch.cuda.synchronize()
        fn = ctypes.CDLL(LIB).sp_bb32_debug_read
        buf = (ctypes.c_ulonglong * (256 * 8 * 10))()
        assert fn(buf, 256 * 8 * 10) == 0
        d = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 10).astype(np.float64)
        names = ["prologue", "conv1 MFMA loops", "conv1 epilogues", "conv2 MFMA loops", "conv2 epilogues + stores", "barrier after conv1", "wait + barrier at the end", "halo requests",
                 "lifetime", "lifetime (10 ns ticks)"]
        print("per wave, mean over workgroups (s_memtime ticks):")
        for k, n in enumerate(names):
            print(f"  {n:24s} " + " ".join(f"{d[:, w, k].mean():9.0f}" for w in range(8)))
        life, rt = d[:, :, 8].mean(), d[:, :, 9].mean()
        print(f"  s_memtime ticks per us: {life / (rt / 100.0):.1f}; lifetime {rt / 100.0:.2f} us")
    same = torch.equal(outs[True].view(torch.int16), outs[False].view(torch.int16))
    mb = B * H * W * CH * 2 / 1e6
    print(f"BasicBlock c{CH} bs={B} {H}x{W}: fused {res[True]:.1f} us ({2 * mb / res[True]:.2f} TB/s of {2 * mb:.0f} MB), two convs {res[False]:.1f} us, "
          f"bit-identical: {same} (SP_BB32_W8={os.environ.get('SP_BB32_W8', 'default')})")


if __name__ == "__main__":
    main()
