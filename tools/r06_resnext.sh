#!/bin/bash
# Round 6: the grouped nets (resnext50_32x4d, not a BASELINE config) measured once: inference bf16 / fp32, train step bf16 / fp32, rocprofv3 kernel statistics
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_resnext
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --backbone resnext50_32x4d --no-cpu-baseline"
for dt in bf16 f32; do
  timeout 300 $B --dtype $dt --steps 20 --warmup 5 --tiles $OUT/infer_${dt}_tiles.json > $OUT/infer_$dt.json 2> $OUT/infer_$dt.err || { tail -5 $OUT/infer_$dt.err; exit 1; }
  timeout 300 $B --mode train --dtype $dt --batch 32 --steps 20 --warmup 5 --tiles $OUT/train_${dt}_tiles.json > $OUT/train_$dt.json 2> $OUT/train_$dt.err || { tail -5 $OUT/train_$dt.err; exit 1; }
  echo "$dt: infer $(grep -o '"value": [0-9.]*' $OUT/infer_$dt.json | head -1) train $(grep -o '"value": [0-9.]*' $OUT/train_$dt.json | head -1)"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_infer -- $B --dtype bf16 --steps 10 --warmup 3 --interleave 1 --tiles $OUT/infer_bf16_tiles.json --no-kernel-events > $OUT/trace_infer.log 2>&1 || echo "infer trace failed"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- $B --mode train --dtype bf16 --batch 32 --steps 10 --warmup 3 --tiles $OUT/train_bf16_tiles.json --no-kernel-events > $OUT/trace_train.log 2>&1 || echo "train trace failed"
find $OUT -name "*_kernel_trace.csv" -size +4M -delete
find $OUT -name "*.db" -delete
for t in infer train; do f=$(find $OUT/trace_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_bf16_kernel_stats.csv && head -12 $f; done
