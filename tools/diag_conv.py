"""Diagnostic (not shipped): per-segment cycle breakdown of one conv launch from the SP_DIAG build of the library.
    SIMPLE_POSE_HIP_LIB=simple_pose_amd/lib/libsimple_pose_hip_diag.so python tools/diag_conv.py
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_pose_amd import _lib, engine, synth

lib = _lib.lib()
dev = "cuda:0"
def run(name, B, cin, h, w, cout, k, s, p, deconv=False):
    global NK
    NK = (4 * cin if deconv else cin * k * k) // 32
    b = engine.ProgramBuilder(h, w); b.p.shapes["input"] = (h, w, cin)
    if deconv:
        wt = torch.randn(cin, cout, 4, 4, device=dev) * 0.02
        out = b.deconv_k4s2p1("input", wt, relu=True)
    else:
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.02
        out = b.conv("input", wt, stride=s, pad=p, relu=True)
    prog = b.p; prog.out_name = out; prog.out_shape = prog.shapes[out]
    x = torch.randn(B, h, w, cin, device=dev)
    for _ in range(3): prog.run(x)
    torch.cuda.synchronize()
    lib.sp_debug_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); prog.run(x); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4096 * 4 * 12))()
    lib.sp_debug_read(buf, 4096 * 4 * 12)
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4, 12)
    d = raw.astype(np.float64)
    keep = raw[:, 0, 7] > 0
    raw = raw[keep]; d = d[keep]
    rt = (raw[:, :, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)  # 100 MHz ticks over the block's life
    rt0 = (raw[:, 0, 7] >> np.uint64(32)).astype(np.float64)
    nk = round(float(np.median(d[:, :, 1] / 1100.0))) if False else None
    tiles = np.full_like(d[:, :, 7:8], NK)
    per = (d[:, :, :7] / tiles).mean(axis=(0, 1))
    names = ["step0(loads+16mfma)", "step1", "step2", "ds_write", "step3a(8mfma)", "barrier", "step3b(frag+8mfma)"]
    print(f"{name}: {e0.elapsed_time(e1)*1e3:.1f} us, blocks sampled {len(d)}, tiles/wave {tiles.mean():.1f}, cycles per K-tile per wave = {per.sum():.0f}")
    for n, v in zip(names, per): print(f"    {n:22s} {v:8.1f}")
    print("    per-wave totals (first block):", (d[0, :, :7].sum(-1) / d[0, :, 7]).round(0))
    print(f"    prologue {d[:,:,8].mean():.0f}  loop {d[:,:,9].mean():.0f}  epilogue(incl. store drain) {d[:,:,10].mean():.0f} cycles;  block lifetime {(d[:,:,8]+d[:,:,9]+d[:,:,10]).mean():.0f}")
    life = (d[:,:,8]+d[:,:,9]+d[:,:,10])
    print(f"    in-kernel clock = {(life / rt).mean() * 0.1:.3f} GHz;  first->last block entry spread {(rt0.max()-rt0.min())*0.01:.1f} us (wraps at 42 s)")
    print(f"    of the epilogue, waiting for the stores to drain (vmcnt(0)): {d[:,:,11].mean():.0f} cycles")

run("deconv6-like 256->256 k4s2 64x48... (B=128, 32x24)", 128, 256, 32, 24, 256, 4, 2, 1, deconv=True)
run("layer3 conv2 3x3 256 (B=128,16x12)", 128, 256, 16, 12, 256, 3, 1, 1)
run("layer2 conv1 1x1 512->128 (B=128, 32x24)", 128, 512, 32, 24, 128, 1, 1, 0)
