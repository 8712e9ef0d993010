"""Timeline summary of ONE train step from a rocprofv3 --kernel-trace CSV of `bench.py --mode train`:
    python tools/step_timeline.py <kernel_trace.csv> [--step -1] [--list]
The last steps of the run are found through the per-step `mse_partial_kernel` launch; per HIP stream (Queue_Id): busy time, idle gaps,
launch count; per kernel family: launches and summed duration.  --list prints every launch of the step in start order."""
import argparse
import collections
import csv
import re


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-2, help="which step (index into the list of steps found; default: second to last)")
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.csv)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    marks = [i for i, r in enumerate(rows) if "nchw_to_nhwc8_bf16_kernel" in r["Kernel_Name"] or "nchw_to_nhwc4_kernel" in r["Kernel_Name"]]
    lo, hi = marks[args.step], marks[args.step + 1]
    step = rows[lo:hi]
    t0, t1 = step[0]["s"], max(r["e"] for r in step)
    print(f"step: {len(step)} launches, {(t1 - t0) / 1e3:.1f} us from first start to last end")
    by_q = collections.defaultdict(list)
    for r in step:
        by_q[r["Queue_Id"]].append(r)
    for q, rs in sorted(by_q.items()):
        busy = sum(r["e"] - r["s"] for r in rs)
        print(f"  queue {q}: {len(rs)} launches, busy {busy / 1e3:.1f} us, first {(rs[0]['s'] - t0) / 1e3:.1f} last end {(max(r['e'] for r in rs) - t0) / 1e3:.1f}")
    fam = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = short(r["Kernel_Name"])
        k = re.sub(r"<.*", "<>", k) if k.startswith("conv_igemm") else k
        fam[(r["Queue_Id"], k)][0] += 1
        fam[(r["Queue_Id"], k)][1] += r["e"] - r["s"]
    for (q, k), (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  q{q} {k[:70]:70s} {n:4d} x  {t / 1e3:8.1f} us")
    if args.list:
        prev = {}
        for r in step:
            q = r["Queue_Id"]
            gap = (r["s"] - prev[q]) / 1e3 if q in prev else 0.0
            prev[q] = r["e"]
            print(f"{(r['s'] - t0) / 1e3:9.1f} q{q} {(r['e'] - r['s']) / 1e3:7.1f} us gap {gap:6.1f}  {short(r['Kernel_Name'])[:80]}")


if __name__ == "__main__":
    main()
