#!/usr/bin/env python3
"""Turn the rocprofv3 CSV output of tools/run_profiles.sh (gpurun_out/prof_<tag>) into the committed summaries under profiles/.

    python tools/profile_summary.py gpurun_out/prof_r02 profiles/r02

per config <cfg> = <arch>_<dtype> (dconv_f32, dconv_bf16, duc_bf16, hrnet_w32_bf16):
        profiles/r02_<cfg>_kernel_stats.csv   verbatim --kernel-trace --stats summary of `bench.py --arch .. --dtype ..`
        profiles/r02_<cfg>_traffic.json       per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch and the HBM bytes per launch
                                              with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts the 128-B
                                              requests of wide coalesced reads as 64 B -> x2; units are KiB)
        profiles/r02_<cfg>_bench.json         the unprofiled bench line of the same config
        profiles/r02_<cfg>_tiles.json         the tile table the profiled runs were pinned to
plus profiles/r02_train_bf16_kernel_stats.csv / _bench.json and profiles/r02_summary.md
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    m = re.search(r"(conv_\w+_kernel<[^>]*>|\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def agg(path, cname):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = short(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def first(pattern):
    """Newest match (gpurun merges every call's output into gpurun_out/, so older collections' files linger next to the new ones)."""
    found = glob.glob(pattern, recursive=True)
    return max(found, key=os.path.getmtime) if found else None


def bench_line(path):
    if path and os.path.isfile(path):
        for line in open(path):
            if line.startswith('{"metric"'):
                return json.loads(line)
    return None


def one_config(src, dst, cfg, fh):
    d = os.path.join(src, cfg)
    stats = first(os.path.join(d, "trace", "**", "*_kernel_stats.csv"))
    if not stats:
        return
    shutil.copy(stats, f"{dst}_{cfg}_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    traffic = {}
    fpath, wpath = first(os.path.join(d, "fetch", "**", "*_counter_collection.csv")), first(os.path.join(d, "write", "**", "*_counter_collection.csv"))
    if fpath and wpath:
        f, w = agg(fpath, "FETCH_SIZE"), agg(wpath, "WRITE_SIZE")
        for k, (n, fs) in f.items():
            wn, ws = w.get(k, [n, 0.0])
            traffic[k] = {"launches_profiled": n, "fetch_size_kib_per_launch": round(fs / n, 1), "write_size_kib_per_launch": round(ws / max(wn, 1), 1),
                          "hbm_bytes_per_launch": int((2.0 * fs / n + ws / max(wn, 1)) * 1024)}
        with open(f"{dst}_{cfg}_traffic.json", "w") as out:
            json.dump(traffic, out, indent=1, sort_keys=True)
    if os.path.isfile(os.path.join(d, "tiles_bs128.json")) and not os.path.isfile(f"{dst}_{cfg}_tiles.json"):
        shutil.copy(os.path.join(d, "tiles_bs128.json"), f"{dst}_{cfg}_tiles.json")
    line = bench_line(os.path.join(d, "bench_unprofiled.json"))
    if line:
        r = line.get("roofline") or {}
        if r and r.get("traffic") is None and r.get("kernel") in traffic:
            # the bench run preceded this summary (the table it looked in was the previous collection's): join the same collection's PMC
            # figure for the kernel it named
            r["traffic"] = traffic[r["kernel"]]["hbm_bytes_per_launch"]
            r["traffic_source"] = "joined by tools/profile_summary.py from the FETCH_SIZE / WRITE_SIZE passes of the same collection"
        with open(f"{dst}_{cfg}_bench.json", "w") as out:
            out.write(json.dumps(line) + "\n")
    prof = bench_line(os.path.join(d, "bench_trace.log"))
    fh.write(f"\n## {cfg}: `python3 bench.py --arch {cfg.rsplit('_', 1)[0]} --dtype {cfg.rsplit('_', 1)[1]} --steps 10 --warmup 3 --interleave 1` under rocprofv3 --kernel-trace --stats\n\n")
    fh.write("(kernel statistics and counters: ONE batch in flight, so that a kernel's duration means what the roofline's HIP events mean; the unprofiled "
             "line is the default command - engine.InterleavedForward: two batches in flight on independent streams on the ResNets, three single-stream forwards on HRNet)\n\n")
    if line:
        r = line.get("roofline") or {}
        fh.write(f"unprofiled: **{line['value']} img/s**, {line['ms_per_step']} ms/step; dominant kernel `{r.get('kernel')}` "
                 f"{r.get('achieved')} {r.get('unit')} = {r.get('frac')} of {r.get('peak')}, avg launch {r.get('avg_launch_us')} us (HIP events), "
                 f"traffic {r.get('traffic')} B/launch; all conv launches {((r.get('all_conv_kernels') or {}).get('frac'))} of peak\n\n")
    if prof:
        fh.write(f"profiled run (profiling perturbs clocks): {prof['value']} img/s, {prof['ms_per_step']} ms/step\n\n")
    fh.write("| kernel | calls | avg us | % GPU time | HBM MB / launch (PMC, corrected) |\n|---|---|---|---|---|\n")
    for r in rows[:14]:
        k = short(r["Name"])
        t = traffic.get(k, {}).get("hbm_bytes_per_launch")
        fh.write(f"| `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | {'' if t is None else round(t / 1e6, 1)} |\n")


def train(src, dst, fh):
    for dt in ("bf16", "f32"):
        d = os.path.join(src, f"train_{dt}")
        stats = first(os.path.join(d, "trace", "**", "*_kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats, f"{dst}_train_{dt}_kernel_stats.csv")
        if os.path.isfile(os.path.join(d, "tiles_bs32.json")) and not os.path.isfile(f"{dst}_train_{dt}_tiles.json"):
            shutil.copy(os.path.join(d, "tiles_bs32.json"), f"{dst}_train_{dt}_tiles.json")
        rows = list(csv.DictReader(open(stats)))
        traffic = {}
        fpath, wpath = first(os.path.join(d, "fetch", "**", "*_counter_collection.csv")), first(os.path.join(d, "write", "**", "*_counter_collection.csv"))
        if fpath and wpath:
            f, w = agg(fpath, "FETCH_SIZE"), agg(wpath, "WRITE_SIZE")
            for k, (n, fs) in f.items():
                wn, ws = w.get(k, [n, 0.0])
                traffic[k] = {"launches_profiled": n, "fetch_size_kib_per_launch": round(fs / n, 1), "write_size_kib_per_launch": round(ws / max(wn, 1), 1),
                              "hbm_bytes_per_launch": int((2.0 * fs / n + ws / max(wn, 1)) * 1024)}
            with open(f"{dst}_train_{dt}_traffic.json", "w") as out:
                json.dump(traffic, out, indent=1, sort_keys=True)
        steps = 3 + 10 + 5            # warm-up + timed + the bench's step-split steps (no kernel-event steps: --no-kernel-events)
        fh.write(f"\n## train step, {dt} compute, 32 images (`python3 bench.py --mode train --dtype {dt} --batch 32 --steps 10 --warmup 3 --no-kernel-events --tiles <tracked>`)\n\n")
        step_kernels = [r for r in rows if "copyBuffer" not in r["Name"] and "pack_" not in r["Name"] and "at::native" not in r["Name"].split("<")[0]]
        calls = sum(int(r["Calls"]) for r in rows)
        fh.write(f"{calls} kernel launches in the profiled run of {steps} steps (tile table pinned: no tuner launches; one-time weight upload / "
                 f"packing included) = **{calls / steps:.0f} per step**\n\n")
        fh.write("| kernel | calls | per step | avg us | % GPU time | HBM MB / launch (PMC, corrected) |\n|---|---|---|---|---|---|\n")
        for r in rows[:24]:
            k = short(r["Name"])
            t = traffic.get(k, {}).get("hbm_bytes_per_launch")
            fh.write(f"| `{k}` | {r['Calls']} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | "
                     f"{'' if t is None else round(t / 1e6, 1)} |\n")
        if traffic:
            tot = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in traffic.values()) / steps
            fh.write(f"\nHBM bytes per step, all kernels (PMC, corrected): **{tot / 1e9:.2f} GB** = {tot / 32 / 1e6:.0f} MB per image\n")
        line = bench_line(os.path.join(d, "bench_unprofiled.json"))
        if line:
            r = line.get("roofline") or {}
            if r and r.get("traffic") is None and traffic:
                name = r.get("kernel", "")
                parts = [traffic.get(k, {}).get("hbm_bytes_per_launch") for k in name.split(" + ")]
                if parts and all(p is not None for p in parts):
                    r["traffic"] = int(sum(parts))
                elif "launches)" in name:          # a family of conv_igemm instantiations: forward = STATS epilogue (8th flag), dgrad = not
                    want, num, den = "forward" in name, 0.0, 0.0
                    for k, v in traffic.items():
                        m = re.match(r"conv_igemm_kernel<(.*)>", k)
                        fl = [t.strip() for t in m.group(1).split(",")] if m else []
                        if len(fl) >= 8 and (fl[7] == "true") == want:
                            num += v["hbm_bytes_per_launch"] * v["launches_profiled"]; den += v["launches_profiled"]
                    if den:
                        r["traffic"] = int(num / den)
                if r.get("traffic") is not None:
                    r["traffic_source"] = "joined by tools/profile_summary.py from the FETCH_SIZE / WRITE_SIZE passes of the same collection"
            with open(f"{dst}_train_{dt}_bench.json", "w") as out:
                out.write(json.dumps(line) + "\n")
            fh.write(f"\n`train_{dt}`: {line['value']} img/s, {line['ms_per_step']} ms/step, host enqueue {line.get('host_enqueue_ms_per_step')} ms/step, "
                     f"split {line.get('step_split_ms')}, roofline {json.dumps(line.get('roofline'))}\n")


def micro(src, dst, fh):
    d = os.path.join(src, "micro")
    stats = first(os.path.join(d, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, f"{dst}_micro_kernel_stats.csv")
    for ext in ("json", "md"):
        if os.path.isfile(os.path.join(d, f"table.{ext}")):
            shutil.copy(os.path.join(d, f"table.{ext}"), f"{dst}_micro_table.{ext}")
    if os.path.isfile(os.path.join(d, "table.md")):
        fh.write("\n## HBM-bound kernels one by one (`python3 tools/bench_micro.py`, HIP events; rocprofv3 durations of the same run: "
                 f"`{os.path.basename(dst)}_micro_kernel_stats.csv`)\n\n")
        fh.write(open(os.path.join(d, "table.md")).read())


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + "_summary.md", "w") as fh:
        fh.write("# rocprofv3 summaries (1 x MI355X; tools/run_profiles.sh -> tools/profile_summary.py)\n")
        for cfg in ("dconv_f32", "dconv_bf16", "duc_bf16", "hrnet_w32_bf16"):
            one_config(src, dst, cfg, fh)
        train(src, dst, fh)
        micro(src, dst, fh)
