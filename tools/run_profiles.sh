#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/run_profiles.sh r01        -> gpurun_out/prof_r01/{trace,fetch,write,train_bf16}/...
# then, back in the build container:     python tools/profile_summary.py gpurun_out/prof_r01 profiles/r01
# Kernel trace and counters are separate passes (a --pmc pass never carries a trace domain).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
TILES=$ROOT/profiles/${TAG}_tiles_bs128.json
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --tiles $TILES"
# untimed: make sure the tile table exists so that the tuner's trial launches stay out of the statistics
[ -f $TILES ] || python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --tiles $TILES > /dev/null 2>&1
cp $TILES $OUT/ 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH --no-kernel-events > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --no-kernel-events > $OUT/bench_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bf16 -- python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_train_bf16.log 2>&1
timeout 300 python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
timeout 300 python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train_bf16.json 2>/dev/null
timeout 300 python3 $ROOT/bench.py --mode train --dtype f32 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_train_f32.json 2>/dev/null
# keep only the summaries (the raw per-dispatch traces are large)
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
