#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/<dir>) into the committed summaries under profiles/.

    python tools/profile_summary.py gpurun_out/prof_r1 profiles/r01

writes  profiles/r01_kernel_stats.csv   (verbatim --kernel-trace --stats summary)
        profiles/r01_traffic.json       (per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch and the HBM bytes
                                         per launch with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts
                                         128-B requests as 64 B for wide coalesced reads -> x2; units are KiB)
        profiles/r01_summary.md
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
    m = re.search(r"(conv_\w+_kernel<[^>]*>|\w+_kernel(?:<\d+>)?)", name)
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


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, dst + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    f = agg(glob.glob(os.path.join(src, "fetch", "**", "*_counter_collection.csv"), recursive=True)[0], "FETCH_SIZE")
    w = agg(glob.glob(os.path.join(src, "write", "**", "*_counter_collection.csv"), recursive=True)[0], "WRITE_SIZE")
    traffic = {}
    for k, (n, fs) in f.items():
        ws = w.get(k, [n, 0.0])[1]
        traffic[k] = {"launches_profiled": n, "fetch_size_kib_per_launch": round(fs / n, 1),
                      "write_size_kib_per_launch": round(ws / max(w.get(k, [n])[0], 1), 1),
                      "hbm_bytes_per_launch": int((2.0 * fs / n + ws / max(w.get(k, [n])[0], 1)) * 1024)}
    with open(dst + "_traffic.json", "w") as fh:
        json.dump(traffic, fh, indent=1, sort_keys=True)
    with open(dst + "_summary.md", "w") as fh:
        fh.write("# rocprofv3 summary (python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline, 1 x MI355X)\n\n")
        fh.write("| kernel | calls | avg us | % GPU time | HBM MB / launch (PMC, corrected) |\n|---|---|---|---|---|\n")
        for r in rows[:12]:
            k = short(r["Name"])
            t = traffic.get(k, {}).get("hbm_bytes_per_launch")
            fh.write(f"| `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | "
                     f"{'' if t is None else round(t / 1e6, 1)} |\n")
        log = os.path.join(src, "bench_trace.log")
        if os.path.isfile(log):
            for line in open(log):
                if line.startswith('{"metric"'):
                    fh.write("\nbench line of the profiled run (profiling perturbs clocks; see BENCH for the unprofiled number):\n\n```\n" + line.strip() + "\n```\n")


def train_summary(src, dst):
    """profiles/<tag>_train_bf16_kernel_stats.csv + a short table appended to the summary."""
    found = glob.glob(os.path.join(src, "train_bf16", "**", "*_kernel_stats.csv"), recursive=True)
    if not found:
        return
    shutil.copy(found[0], dst + "_train_bf16_kernel_stats.csv")
    rows = list(csv.DictReader(open(found[0])))
    with open(dst + "_summary.md", "a") as fh:
        fh.write("\n## train step, bf16 compute, 32 images (python3 bench.py --mode train --dtype bf16 --batch 32 --steps 10 --warmup 3)\n\n")
        fh.write("| kernel | calls | avg us | % GPU time |\n|---|---|---|---|\n")
        for r in rows[:16]:
            fh.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        for name in ("bench_train_bf16.json", "bench_train_f32.json"):
            path = os.path.join(src, name)
            if os.path.isfile(path):
                for line in open(path):
                    if line.startswith('{"metric"'):
                        shutil.copy(path, dst + "_" + name)
                        d = json.loads(line)
                        fh.write(f"\n`{name}`: {d['value']} img/s, {d['ms_per_step']} ms/step, split {d.get('step_split_ms')}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    train_summary(sys.argv[1], sys.argv[2])
    for extra in ("bench_unprofiled.json",):
        if os.path.isfile(os.path.join(sys.argv[1], extra)):
            shutil.copy(os.path.join(sys.argv[1], extra), sys.argv[2] + "_" + extra)
