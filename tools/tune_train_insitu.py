"""In-situ tile choice for the 32-image train step (bf16): every forward / dgrad launch family timed INSIDE the running step (HIP events on its
stream: weight gradients, optimizer and the shortcut branch running beside it as they do in the step) under seven candidate tables - the tracked
one and "tile X wherever legal" for the six tiles - then per launch family the fastest candidate, and the resulting table A/B'ed against the
tracked one.  `PoseTrainer.autotune` times launches alone, back to back on a warm cache: at 32 images its choices are inside its own noise.

    python tools/tune_train_insitu.py --out gpurun_out/tiles_insitu.json [--dtype bf16]
"""
import argparse, json, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=12)
    args = ap.parse_args()
    from simple_pose_amd import _lib, synth
    from simple_pose_amd.nets import pose_resnet_dconv
    from simple_pose_amd.commons.transforms import RefineSimpleTransform
    from simple_pose_amd.train import PoseTrainer
    from oracle import nets_oracle
    dev, B = torch.device("cuda", 0), 32
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv")
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=0).items()}
    model = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    model.load_state_dict(sd, strict=True)
    model.to(dev).train()
    tr = PoseTrainer(model, lr=1e-3, dtype="bf16" if args.dtype == "bf16" else "fp32")
    tracked = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_train_{args.dtype}_tiles.json")))[-1]))
    base = synth.input_images(8, seed=100)
    x = torch.from_numpy(np.concatenate([base] * 4, 0)[:B]).to(dev)
    joints = torch.from_numpy(synth.joints_batch(B, 17, seed=200)).to(dev)
    targets, mask = RefineSimpleTransform.get_heat_map(joints, 2.0, (48, 64))

    def table_for(tile):
        t = dict(tracked)
        if tile is None:
            return t
        tm, tn = tile
        for name, layer in tr.layers.items():
            if layer.d_fwd.n_pad % tn == 0:
                t[name] = [tm, tn]
            if layer.need_dgrad:
                ok = all(d.n_pad % tn == 0 for d in layer.d_dgrad)
                for i, d in enumerate(layer.d_dgrad):
                    if ok:
                        t[f"{name}.dgrad{i}"] = [tm, tn]
        return t

    def families(tab):
        """per launch family the tile it runs under `tab` (a dgrad family = all its phases)"""
        out = {}
        for name, layer in tr.layers.items():
            out[("forward", name)] = tuple(tab.get(name, [0, 0]))
            if layer.need_dgrad:
                out[("dgrad", name)] = tuple(tuple(tab.get(f"{name}.dgrad{i}", [0, 0])) for i in range(len(layer.d_dgrad)))
        return out

    def step_ms(tab, n=20):
        tr.set_tiles(tab, B)
        for _ in range(4):
            tr.step(x, targets, mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            tr.step(x, targets, mask)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n

    def per_family_us(tab):
        tr.set_tiles(tab, B)
        for _ in range(3):
            tr.step(x, targets, mask)
        tr.kernel_events = []
        for _ in range(args.steps):
            tr.step(x, targets, mask)
        torch.cuda.synchronize()
        ev, tr.kernel_events = tr.kernel_events, None
        acc = {}
        for kind, name, flops, a, b in ev:
            if kind in ("forward", "dgrad"):
                q = acc.setdefault((kind, name), [0.0, 0])
                q[0] += 1e3 * a.elapsed_time(b); q[1] += 1
        return {k: v[0] / v[1] for k, v in acc.items()}

    cands = [None] + list(_lib.CONV_TILES)
    tabs = [table_for(c) for c in cands]
    times = []
    for c, tab in zip(cands, tabs):
        try:
            times.append(per_family_us(tab))
        except Exception as e:                       # a tile some layer cannot run: candidate dropped
            print("candidate", c, "failed:", str(e)[:120], flush=True)
            times.append(None)
        else:
            print("candidate", c, "sum of conv families %.1f us" % sum(times[-1].values()), flush=True)
    best = dict(tracked)
    moved = 0
    for key in times[0]:
        kind, name = key
        opts = [(t[key], i) for i, t in enumerate(times) if t is not None and key in t]
        us, i = min(opts)
        if i != 0 and us < 0.97 * times[0][key]:           # move only for a clear win (3 %)
            moved += 1
            if kind == "forward":
                best[name] = tabs[i][name]
            else:
                for j in range(len(tr.layers[name].d_dgrad)):
                    best[f"{name}.dgrad{j}"] = tabs[i][f"{name}.dgrad{j}"]
    print("launch families moved:", moved, "of", len(times[0]), flush=True)
    for rep in range(3):
        print("step ms: tracked %.3f  in-situ %.3f" % (step_ms(tracked), step_ms(best)), flush=True)
    with open(args.out, "w") as fh:
        json.dump(best, fh)


if __name__ == "__main__":
    main()
