#!/usr/bin/env python3
"""Per-layer comparison of every way a conv launch can run (tools, not product): for each distinct conv shape of a network at
batch B, every legal (tile, kernel) candidate is (1) checked bit for bit against the first candidate's output and (2) timed
(median of `--rounds` timings of `--reps` back-to-back launches, HIP events on the launch stream).

    python tools/bench_conv_layers.py --arch dconv --dtype bf16 --batch 128 --out gpurun_out/layers_ring_dconv.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="dconv", choices=["dconv", "duc", "hrnet_w32"])
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="substring filter on layer names")
    args = ap.parse_args()

    import numpy as np
    import torch

    from simple_pose_amd import _lib, synth
    from simple_pose_amd._lib import ConvDesc
    from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc

    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    if args.arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        model = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(model.cfg, 17), seed=0)
    else:
        mod = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[args.arch]
        model = mod.resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict([(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()], seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    if args.dtype == "bf16":
        model.compute_dtype = "bf16"
    B = args.batch
    base = synth.input_images(8, seed=100)
    x = torch.from_numpy(np.concatenate([base] * ((B + 7) // 8), 0)[:B]).to(dev)
    prog = model.hip_program(x)
    prog.multi_stream = False
    prog.run(x)
    torch.cuda.synchronize()
    bufs = dict(prog._alloc(B, dev))
    bufs["input"] = x
    bufs[prog.out_name] = torch.empty((B,) + tuple(prog.out_shape), dtype=torch.float32, device=dev)
    stream = _lib.current_stream()
    P = _lib.ptr
    rows, seen = [], {}
    bad = 0
    for op in prog.ops:
        if op.kind != "conv" or (args.only and not any(t in op.name for t in args.only.split(","))):
            continue
        d = op.desc
        d.batch = B
        key = tuple(getattr(d, f) for f, _ in ConvDesc._fields_ if f not in ("tile_m", "tile_n", "kernel")) + (op.res is not None,)
        if key in seen:
            rows.append(dict(seen[key], layer=op.name, dup=True))
            continue
        keep = (d.tile_m, d.tile_n, d.kernel, op.direct)
        ref = None
        res = {}
        for cand in prog._candidates(lib, op):
            direct = cand[0] < 0
            if not direct:
                d.tile_m, d.tile_n, d.kernel = cand
            fn = lib.sp_conv3x3_direct if direct else lib.sp_conv2d_fwd
            a = (d, P(bufs[op.src]), P(op.w), P(op.scale), P(op.shift), P(bufs[op.res]) if op.res else None, P(bufs[op.dst]), stream)
            bufs[op.dst].zero_()
            _lib.check(fn(*a), op.name)
            torch.cuda.synchronize()
            out = bufs[op.dst].view(torch.int16 if bufs[op.dst].dtype == torch.bfloat16 else torch.int32).clone()
            if ref is None:
                ref = out
                same = True
            else:
                same = bool(torch.equal(ref, out))
                if not same:
                    bad += 1
                    nd = int((ref != out).sum().item())
                    print(f"  MISMATCH {op.name} {cand}: {nd} of {out.numel()} elements differ", flush=True)
            ts = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn(*a)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / args.reps * 1e3)
            us = sorted(ts)[len(ts) // 2]
            res["%dx%d/k%d" % cand] = {"us": round(us, 1), "tflops": round(op.flops * B / us / 1e6, 1), "same_bits": same}
        d.tile_m, d.tile_n, d.kernel, op.direct = keep
        best = min(res, key=lambda k: res[k]["us"])
        best_old = min((k for k in res if k.endswith("k0")), key=lambda k: res[k]["us"])
        row = {"layer": op.name, "gflop": round(op.flops * B / 1e9, 2), "M": B * d.grid_h * d.grid_w, "N": d.c_out, "K": d.k_pad,
               "phases": d.phases_y * d.phases_x, "best": best, "best_us": res[best]["us"], "best_igemm": best_old,
               "best_igemm_us": res[best_old]["us"], "cands": res}
        seen[key] = row
        rows.append(row)
        ring = {k: v["us"] for k, v in res.items() if k.endswith("k1") or k.endswith("k2")}
        print(f"{op.name:26s} M={row['M']:7d} N={row['N']:5d} K={row['K']:5d}  igemm {best_old:>11s} {row['best_igemm_us']:7.1f} us | "
              + " ".join(f"{k[:-3] + ('(pw)' if k.endswith('k2') else '')}:{v:.1f}" for k, v in ring.items()) + f" | best {best} {res[best]['tflops']:.0f} TF", flush=True)
    tot_old = sum(r["best_igemm_us"] for r in rows)
    tot_new = sum(r["best_us"] for r in rows)
    print(f"sum over {len(rows)} conv launches: best igemm {tot_old:.0f} us, best of all {tot_new:.0f} us; mismatching candidates: {bad}")
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rows, fh, indent=0)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
