"""Throughput of the HBM-bound kernels of the path ("HBM roofline for decode / encode / Adam / BN") at BASELINE sizes,
operands resident in HBM, each kernel timed alone with HIP events (median of `--rounds` x `--reps` back-to-back launches):

    python tools/bench_micro.py [--out table.json] [--md table.md]          (also: python bench.py --mode micro)

Algorithmic bytes = every operand read once + every result written once; achieved TB/s against the 8 TB/s HBM3E spec and the
6.3 TB/s a float4 copy reaches on this chip (MI355X_MICROARCH.md).  Buffers far below the 256 MB Infinity Cache are served from it
when a launch is repeated back to back - those rows say so (`cache_resident`).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

HBM_SPEC_TBS, HBM_COPY_TBS = 8.0, 6.3


def main(argv=None, emit=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--md", default=None)
    args = ap.parse_args(argv)
    from simple_pose_amd import _lib, synth

    lib, P, dev = _lib.lib(), _lib.ptr, "cuda:0"
    st = _lib.current_stream()
    rows = []

    def timed(name, replaces, nbytes, fn, note=""):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / args.reps)
        us = sorted(ts)[len(ts) // 2]
        tbs = nbytes / us / 1e6
        rows.append({"kernel": name, "replaces": replaces, "MB": round(nbytes / 1e6, 1), "us": round(us, 1), "TBps": round(tbs, 2),
                     "frac_of_8TBps": round(tbs / HBM_SPEC_TBS, 3), "frac_of_copy_rate": round(tbs / HBM_COPY_TBS, 3),
                     "cache_resident": nbytes < 200e6, "note": note})
        print(f"{name:46s} {nbytes / 1e6:8.1f} MB {us:8.1f} us {tbs:6.2f} TB/s  {note}", flush=True)

    # ---- encoder: RefineSimpleTransform.get_heat_map, bs = 128 (commons/transforms.py:167-191) ----
    B, J, H, W = 128, 17, 64, 48
    joints = torch.from_numpy(synth.joints_batch(B, J, seed=1)).to(dev)
    targets = torch.empty((B, J, H, W), device=dev)
    weights = torch.empty((B, J), device=dev)
    timed("sp_encode_gauss_refine bs=128", "transforms.py:167-191", targets.numel() * 4 + joints.numel() * 4,
          lambda: _lib.check(lib.sp_encode_gauss_refine(P(joints), B, J, H, W, 2.0, P(targets), P(weights), st)),
          "one exp per pixel in fp64 (the reference's arithmetic): VALU-bound")
    # ---- decoder for comparison (the bench's step already times it) ----
    heat = torch.randn((B, J, H, W), device=dev)
    tinv = torch.from_numpy(synth.trans_inv_batch(B)).to(dev)
    kps, mv = torch.empty((B, J, 2), device=dev), torch.empty((B, J), device=dev)
    timed("sp_decode_gauss_taylor bs=128", "pose_metrics.py:62-107", heat.numel() * 4 + B * J * 12,
          lambda: _lib.check(lib.sp_decode_gauss_taylor(P(heat), P(tinv), B, J, H, W, 11, P(kps), P(mv), st)),
          "121-tap bit-exact blur: VALU-bound (DESIGN 3.2)")
    # ---- loss: 0.5 * MSE(pred * m, target * m) + gradient (ddp...:94,117) ----
    mask = (torch.rand((B, J), device=dev) > 0.2).float()
    grad = torch.empty_like(heat)
    loss, ws = torch.zeros(1, device=dev), torch.empty(4096, dtype=torch.uint8, device=dev)
    timed("sp_masked_mse + grad bs=128", "ddp...:94,117", heat.numel() * 12,
          lambda: _lib.check(lib.sp_masked_mse(P(heat), P(targets), P(mask), B, J, H * W, P(loss), P(grad), P(ws), st)))
    B32 = 32
    timed("sp_masked_mse + grad bs=32", "ddp...:94,117", B32 * J * H * W * 12,
          lambda: _lib.check(lib.sp_masked_mse(P(heat), P(targets), P(mask), B32, J, H * W, P(loss), P(grad), P(ws), st)),
          "config 4's per-GPU batch: 5 MB, launch-latency-bound")
    # ---- Adam on the flat 34.0 M-float buffers (ddp...:70-72): 28 B per parameter ----
    n = 34_000_000 // 4 * 4
    p_, g_, m_, v_ = (torch.randn(n, device=dev) for _ in range(4))
    v_.abs_()
    timed("sp_adam_step 34.0 M parameters", "ddp...:70-72,119", n * 28,
          lambda: _lib.check(lib.sp_adam_step(P(p_), P(g_), P(m_), P(v_), n, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, st)))
    del p_, g_, m_, v_
    # ---- BatchNorm passes on layer1 shapes (pose_resnet_dconv.py:124-131 forward tail and its backward) ----
    for (Bb, C, tag) in ((128, 256, "layer1 block output, bs=128"), (32, 256, "layer1 block output, bs=32"), (32, 64, "layer1 conv1/conv2, bs=32")):
        rows_n = Bb * 64 * 48
        for bf16 in (0, 1):
            adt = torch.bfloat16 if bf16 else torch.float32
            es = 2 if bf16 else 4
            z = torch.randn((rows_n, C), device=dev).to(adt)
            res = torch.randn((rows_n, C), device=dev).to(adt)
            y = torch.empty_like(z)
            mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            timed(f"sp_bn_apply_nhwc (+res, relu) {'bf16' if bf16 else 'fp32'} C={C} {tag}", "pose_resnet_dconv.py:124-131", rows_n * C * es * 3,
                  lambda: _lib.check(lib.sp_bn_apply_nhwc(P(z), bf16, P(mean), P(invstd), P(gamma), P(beta), P(res), P(y), rows_n, C, 1, None, st)))
            dy = torch.randn((rows_n, C), device=dev)
            dz = torch.empty_like(z)
            dres = torch.empty((rows_n, C), device=dev)
            sg, sb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            # reads dy (fp32), relu mask y, z; writes dz (activation dtype) and dres (fp32)
            nb = rows_n * C * (4 + es + es + es + 4)
            timed(f"sp_bn_train_bwd_apply_nhwc {'bf16' if bf16 else 'fp32'} C={C} {tag}", "loss.backward() through :124-131", nb,
                  lambda: _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(dy), bf16, P(y), P(z), P(mean), P(invstd), P(gamma), P(sg), P(sb), rows_n, rows_n, C,
                                                                   P(dz), P(dres), 0, st)))
            del z, res, y, dy, dz, dres
    # ---- max pool 3x3 s2 on the stem output (pose_resnet_dconv.py:162) ----
    for Bb in (128, 32):
        x = torch.randn((Bb, 128, 96, 64), device=dev)
        yq = torch.empty((Bb, 64, 48, 64), device=dev)
        timed(f"sp_maxpool3x3s2_nhwc fp32 bs={Bb}", "pose_resnet_dconv.py:162", x.numel() * 4 + yq.numel() * 4,
              lambda: _lib.check(lib.sp_maxpool3x3s2_nhwc(P(x), P(yq), Bb, 128, 96, 64, st)))
        xb, yb = x.to(torch.bfloat16), yq.to(torch.bfloat16)
        timed(f"sp_maxpool3x3s2_nhwc_bf16 bs={Bb}", "pose_resnet_dconv.py:162", xb.numel() * 2 + yb.numel() * 2,
              lambda: _lib.check(lib.sp_maxpool3x3s2_nhwc_bf16(P(xb), P(yb), Bb, 128, 96, 64, st)))
        del x, yq, xb, yb
    # ---- input layout (the stem's loader format) ----
    x = torch.randn((128, 3, 256, 192), device=dev)
    y4 = torch.empty((128, 256, 192, 4), device=dev)
    timed("sp_nchw_to_nhwc4 bs=128", "input layout", x.numel() * 4 + y4.numel() * 4,
          lambda: _lib.check(lib.sp_nchw_to_nhwc4(P(x), P(y4), 128, 3, 256, 192, st)))
    out = {"hbm_spec_TBps": HBM_SPEC_TBS, "hbm_copy_TBps": HBM_COPY_TBS, "rows": rows}
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    if args.md:
        with open(args.md, "w") as fh:
            fh.write("| kernel | replaces | algorithmic MB | us | TB/s | of 8 TB/s | of 6.3 TB/s | note |\n|---|---|---|---|---|---|---|---|\n")
            for r in rows:
                note = r["note"] + (" (fits the 256 MB Infinity Cache when repeated)" if r["cache_resident"] else "")
                fh.write(f"| `{r['kernel']}` | `{r['replaces']}` | {r['MB']} | {r['us']} | {r['TBps']} | {r['frac_of_8TBps']} | {r['frac_of_copy_rate']} | {note} |\n")
    summary = {"metric": "HBM-bound kernels, achieved TB/s", "rows": len(rows)}
    if emit is None:
        print(json.dumps(summary))
    else:
        emit(summary)                 # bench.py owns the real stdout (its file descriptor 1 is stderr after claim_stdout())
    return 0


if __name__ == "__main__":
    sys.exit(main())
