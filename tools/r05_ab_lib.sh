#!/bin/bash
# Same-box A/B of two builds of the library on one bench.py command line (alternating, N rounds):
#   bash tools/r05_ab_lib.sh <other .so> <tag> <rounds> -- <bench.py arguments>
OTHER=$1; TAG=$2; ROUNDS=$3; shift 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_ab_$TAG
mkdir -p $OUT
for i in $(seq 1 $ROUNDS); do
  SIMPLE_POSE_HIP_LIB=$OTHER python3 $ROOT/bench.py "$@" > $OUT/prev_$i.json 2> $OUT/prev_$i.err || exit 1
  python3 $ROOT/bench.py "$@" > $OUT/new_$i.json 2> $OUT/new_$i.err || exit 1
done
python3 - "$OUT" <<'PY'
import glob, json, sys
out = sys.argv[1]
for tag in ("prev", "new"):
    vals, us = [], []
    for f in sorted(glob.glob(f"{out}/{tag}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        vals.append(d["value"]); us.append((d["roofline"]["kernel"][:40], d["roofline"]["avg_launch_us"]))
    print(tag, "img/s", vals, "dominant kernel", us)
PY
