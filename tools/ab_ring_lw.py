#!/usr/bin/env python3
"""Same-box A/B of the loader-wave ring kernels (SP_CONV_KERNEL_RING_LW) against the tracked tile table: every layer the tracked table runs on an
8-wave ring tile that also exists with loader waves is switched, the whole forward is timed both ways (interleaved rounds), and every switched
layer is timed alone inside the running forward (HIP events around its launch) so that a mixed table (LW only where it wins) can be written.

    python tools/ab_ring_lw.py --arch dconv --out gpurun_out/lw_dconv_bf16.json [--write-table profiles/r05_dconv_bf16_tiles.json]
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
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", required=True)
    ap.add_argument("--write-table", default=None)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    import glob

    import numpy as np
    import torch
    from simple_pose_amd import _lib, synth
    from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc

    dev = torch.device("cuda", 0)
    if a.arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        model = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(model.cfg, 17), seed=0)
    else:
        model = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[a.arch].resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict([(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()], seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    model.compute_dtype = "bf16"
    B = a.batch
    x = torch.from_numpy(np.concatenate([synth.input_images(8, 100)] * ((B + 7) // 8), 0)[:B]).to(dev)
    prog = model.hip_program(x)
    if a.arch == "hrnet_w32":
        prog.multi_stream = False
    tracked_path = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", f"r*_{a.arch}_bf16_tiles.json")))[-1]
    with open(tracked_path) as fh:
        base = {k: tuple(v) for k, v in json.load(fh).items()}
    lw_tiles = set(_lib.RING_LW_TILES)
    lib = _lib.lib()
    ops = {op.name: op for op in prog.ops if op.kind == "conv"}
    allw = dict(base)
    switched = []
    for k, t in base.items():
        if k in ops and len(t) > 2 and t[2] == _lib.SP_CONV_KERNEL_RING and (t[0], t[1]) in lw_tiles:   # (fused blocks have no conv op of their own)
            allw[k] = (t[0], t[1], _lib.SP_CONV_KERNEL_RING_LW)
            switched.append(k)

    def whole(table):
        prog.set_tiles(table, B)
        return min(prog._step_ms(x, 20) for _ in range(3))

    res = {"tracked": tracked_path, "switched_layers": len(switched), "rounds": []}
    for _ in range(a.rounds):
        res["rounds"].append({"tracked_ms": whole(base), "all_lw_ms": whole(allw)})
    # per layer, inside the running forward: events around the launch of that op (Program.run hook: time via the op's own launch)
    per = {}
    for table_name, table in (("ring", base), ("lw", allw)):
        prog.set_tiles(table, B)
        for _ in range(2):
            prog.run(x)
        torch.cuda.synchronize()
        bufs = prog._alloc(B, dev)
        st = _lib.current_stream()
        for name in switched:
            op = ops[name]
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # the layer alone, back to back x 5 (isolated timing: hot inputs; ranks candidates, not absolute)
                e0.record()
                for _ in range(5):
                    prog._launch(lib, op, bufs, B, st)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            per.setdefault(name, {})[table_name] = round(sorted(ts)[1], 2)
    res["per_layer_us_isolated"] = per
    mixed = dict(base)
    for name, v in per.items():
        if v["lw"] < 0.98 * v["ring"]:
            mixed[name] = allw[name]
    res["mixed_layers"] = sum(1 for k in mixed if mixed[k] != base[k])
    for r in res["rounds"]:
        r["mixed_ms"] = whole(mixed)
    best = min(("tracked", "all_lw", "mixed"), key=lambda n: min(r[n + "_ms"] for r in res["rounds"]))
    res["best"] = best
    res["img_per_s"] = {n: round(B / min(r[n + "_ms"] for r in res["rounds"]) * 1e3, 1) for n in ("tracked", "all_lw", "mixed")}
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ("switched_layers", "mixed_layers", "best", "img_per_s")}))
    if a.write_table:
        tab = {"tracked": base, "all_lw": allw, "mixed": mixed}[best]
        with open(a.write_table, "w") as fh:
            json.dump({k: list(v) for k, v in tab.items()}, fh)


if __name__ == "__main__":
    main()
