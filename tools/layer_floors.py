#!/usr/bin/env python3
"""Per-layer floors of the ResNet50-DConv forward at batch B against what bench.py --layers-out measured (tools, not product).

    python tools/layer_floors.py gpurun_out/layers_dconv_bf16.json bf16 [128] > profiles/r02_dconv_bf16_floors.md

floor of a layer = max(algorithmic bytes / HBM_ACHIEVABLE, algorithmic FLOP / MFMA_PRACTICAL): bytes = input + output (+ residual) +
packed weights, one launch per layer (no cross-layer fusion); HBM_ACHIEVABLE = 4.8 TB/s (what the HBM-bound 1x1 convs of layer1 reach),
MFMA_PRACTICAL = 1.0 PFLOP/s bf16 / 135 TFLOP/s fp32 (the best sustained by any layer of this net)."""
import json
import sys


def shapes(B, es):
    sh = {"conv1": es * B * (4 * 256 * 192) + es * B * 64 * 128 * 96}
    h, w, inpl = 64, 48, 64
    for li, (pl, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), 1):
        for bi in range(n):
            s = 2 if (bi == 0 and li > 1) else 1
            p = f"layer{li}.{bi}"
            sh[p + ".conv1"] = es * B * (inpl * h * w + pl * h * w) + es * inpl * pl
            sh[p + ".conv2"] = es * B * (pl * h * w + pl * (h // s) * (w // s)) + es * 9 * pl * pl
            if bi == 0:
                sh[p + ".downsample"] = es * B * (inpl * h * w // (s * s) + 4 * pl * (h // s) * (w // s)) + es * inpl * 4 * pl
            sh[p + ".conv3"] = es * B * (pl * (h // s) * (w // s) + 2 * 4 * pl * (h // s) * (w // s)) + es * pl * 4 * pl
            h, w, inpl = h // s, w // s, 4 * pl
    for idx in (0, 3, 6):
        sh[f"deconv_layers.{idx}"] = es * B * (inpl * h * w + 256 * 4 * h * w) + es * 16 * inpl * 256
        h, w, inpl = 2 * h, 2 * w, 256
    sh["final_layer"] = es * B * 256 * h * w + 4 * B * 17 * h * w
    return sh


def main():
    path, dt = sys.argv[1], sys.argv[2]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    es, hbm, mfma = (2, 4.8e12, 1.0e15) if dt == "bf16" else (4, 4.8e12, 135e12)
    L = json.load(open(path))
    sh = shapes(B, es)
    groups, rows = {}, []
    for l in L:
        by = sh[l["layer"]]
        th, tm = by / hbm * 1e6, l["gflop"] * 1e9 / mfma * 1e6
        g = groups.setdefault(l["layer"].split(".")[0], [0.0, 0.0, 0.0, 0.0])
        g[0] += l["us"]; g[1] += max(th, tm); g[2] += th; g[3] += tm
        rows.append((l["us"] - max(th, tm), l["layer"], l["us"], th, tm, l["kernel"]))
    tot = [sum(g[i] for g in groups.values()) for i in range(4)]
    print(f"# ResNet50-DConv {dt} forward, bs={B}: measured conv launches vs one-launch-per-layer floors\n")
    print(f"`{path}` (bench.py --layers-out: HIP events around every conv launch); floors: bytes / {hbm / 1e12:.1f} TB/s, FLOP / {mfma / 1e12:.0f} TFLOP/s\n")
    print("| group | measured us | floor us | of which HBM-bound sum | MFMA-bound sum |\n|---|---|---|---|---|")
    for k, g in groups.items():
        print(f"| {k} | {g[0]:.0f} | {g[1]:.0f} | {g[2]:.0f} | {g[3]:.0f} |")
    print(f"| **all conv launches** | **{tot[0]:.0f}** | **{tot[1]:.0f}** | {tot[2]:.0f} | {tot[3]:.0f} |")
    print(f"\nfloor ⇒ {B / (tot[1] * 1e-6) / 1e3:.1f} k img/s for the conv launches alone; measured sum ⇒ {B / (tot[0] * 1e-6) / 1e3:.1f} k img/s\n")
    print("largest gaps (measured - floor):\n\n| layer | measured us | HBM floor | MFMA floor | kernel |\n|---|---|---|---|---|")
    for r in sorted(rows, reverse=True)[:12]:
        print(f"| {r[1]} | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.1f} | `{r[5]}` |")


if __name__ == "__main__":
    main()
