"""Per-layer timing of the weight-gradient launches of the train step (sp_conv2d_wgrad), each layer's real shapes at the per-GPU batch,
timed alone with HIP events (median of `--rounds` x `--reps` back-to-back launches).  Development aid for csrc/conv_wgrad.hip:

    python tools/bench_wgrad.py [--dtype bf16|fp32] [--batch 32] [--head dconv] [--out table.json]

A launch timed alone re-reads operands that the previous repetition left in the 256 MB Infinity Cache, so the numbers are an upper
bound on what the kernel does inside the step; the step itself is what bench.py --mode train measures.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--head", default="dconv", choices=["dconv", "duc"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="substring of the layer names to time")
    ap.add_argument("--grouped", type=float, default=0.0, help="also time the layers in backward order as groups of this many GFLOP "
                    "(sp_conv2d_wgrad_batched, as PoseTrainer launches them)")
    ap.add_argument("--skip-single", action="store_true")
    args = ap.parse_args()
    from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc
    from simple_pose_amd.train import PoseTrainer

    dev = "cuda:0"
    torch.manual_seed(0)
    model = (pose_resnet_dconv if args.head == "dconv" else pose_resnet_duc).resnet50(pretrained=False, num_classes=17).to(dev).train()
    tr = PoseTrainer(model, dtype=args.dtype, collectives=False)
    B = args.batch
    adt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    rows, tot_us, tot_fl = [], 0.0, 0.0
    operands = {}
    for name, L in tr.layers.items():
        if args.only and args.only not in name:
            continue
        d = L.d_wgrad
        d.batch = B
        if L.kind == "conv":
            x = torch.randn((B, d.in_h, d.in_w, d.c_in), device=dev).to(adt)
            dz = torch.randn((B, L.oh, L.ow, L.c_out_buf), device=dev).to(adt)
        else:                                         # transposed conv: g = the layer input, a = dy gathered like the dgrad conv
            x = torch.randn((B, L.h, L.w, L.I), device=dev).to(adt)
            dz = torch.randn((B, d.in_h, d.in_w, d.c_in), device=dev).to(adt)
        operands[name] = (x, dz)
    if args.grouped > 0:
        import ctypes
        from simple_pose_amd import _lib
        lib = _lib.lib()
        groups, cur, fl = [], [], 0.0
        for name in reversed(list(operands)):        # backward order
            cur.append(name)
            fl += tr.layers[name].flops * B
            if fl >= args.grouped * 1e9:
                groups.append(cur); cur, fl = [], 0.0
        if cur:
            groups.append(cur)
        calls = []
        for gnames in groups:
            jobs = (_lib.WgradJob * len(gnames))()
            for j, n in zip(jobs, gnames):
                tr.layers[n].wgrad_job(operands[n][0], operands[n][1], B, j)
            need = ctypes.c_int64(0)
            _lib.check(lib.sp_conv2d_wgrad_workspace(jobs, len(gnames), ctypes.byref(need)))
            if need.value > tr.wgrad_ws.numel() * 4:
                tr.wgrad_ws = torch.empty(need.value // 4 + 1024, dtype=torch.float32, device=dev)
            calls.append((jobs, len(gnames), need.value, sum(tr.layers[n].flops * B for n in gnames)))
        st = _lib.current_stream()

        def run_group(c):
            _lib.check(lib.sp_conv2d_wgrad_batched(c[0], c[1], _lib.ptr(tr.wgrad_ws), tr.wgrad_ws.numel() * 4, st))
        for c in calls:
            run_group(c)
        torch.cuda.synchronize()
        gt = []
        for c in calls:
            ts = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run_group(c)
                e1.record(); e1.synchronize()
                ts.append(1e3 * e0.elapsed_time(e1) / args.reps)
            gt.append(sorted(ts)[len(ts) // 2])
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in calls:
                run_group(c)
            e1.record(); e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1))
        allus = sorted(ts)[len(ts) // 2]
        for gnames, c, us in zip(groups, calls, gt):
            print(f"group {gnames[0]} .. {gnames[-1]} ({c[1]} layers, {c[3] / 1e9:.0f} GFLOP, slabs {c[2] / 1e6:.0f} MB): {us:.1f} us = {c[3] / us / 1e6:.0f} TFLOP/s", flush=True)
        totf = sum(c[3] for c in calls)
        print(f"GROUPED {len(calls)} groups back to back: {allus:.0f} us, {totf / allus / 1e6:.1f} TFLOP/s (sum of groups alone {sum(gt):.0f} us)", flush=True)
    for name, L in tr.layers.items():
        if name not in operands or args.skip_single:
            continue
        d = L.d_wgrad
        x, dz = operands[name]
        L.wgrad(x, dz, B)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                L.wgrad(x, dz, B)
            e1.record()
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / args.reps)
        us = sorted(ts)[len(ts) // 2]
        fl = L.flops * B
        M = B * d.grid_h * d.grid_w
        gbytes = dz.numel() * dz.element_size() + x.numel() * x.element_size() + L.wg["n_valid"] * L.wg["s_n"] * 4
        rows.append({"layer": name, "M": M, "N": L.wg["n_valid"], "K": d.k_pad, "us": round(us, 1), "gflop": round(fl / 1e9, 2),
                     "tflops": round(fl / us / 1e6, 1), "min_MB": round(gbytes / 1e6, 1), "GBps_min": round(gbytes / us / 1e3, 0)})
        tot_us += us
        tot_fl += fl
        print(f"{name:28s} M={M:7d} N={L.wg['n_valid']:5d} K={d.k_pad:5d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {gbytes / us / 1e3:6.0f} GB/s of compulsory bytes",
              flush=True)
    if rows:
        print(f"TOTAL {len(rows)} layers: {tot_us:.0f} us, {tot_fl / tot_us / 1e6:.1f} TFLOP/s")
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"dtype": args.dtype, "batch": B, "total_us": round(tot_us, 1), "tflops": round(tot_fl / tot_us / 1e6, 1), "layers": rows}, fh, indent=0)


if __name__ == "__main__":
    main()
