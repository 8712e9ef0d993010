#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/run_profiles.sh r02        -> gpurun_out/prof_r02/<arch>_<dtype>/{trace,fetch,write}/... + train_bf16/ + bench lines
# then, back in the build container:     python tools/profile_summary.py gpurun_out/prof_r02 profiles/r02
# One block per BASELINE config (dconv f32 = the headline, dconv/duc/hrnet_w32 bf16); kernel trace and counters are separate passes
# (a --pmc pass never carries a trace domain); the program itself follows `--` (python3 ..., no wrapper).
set -u
TAG=${1:-r02}
CONFIGS=${2:-"dconv:f32 dconv:bf16 duc:bf16 hrnet_w32:bf16"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $CONFIGS; do
  arch=${cfg%%:*}; dt=${cfg##*:}
  D=$OUT/${arch}_${dt}
  mkdir -p $D
  TILES=$ROOT/profiles/${TAG}_${arch}_${dt}_tiles.json      # the tracked table (what bench.py loads by default); picked if missing
  # kernel statistics and counters with ONE batch in flight (the per-kernel durations then mean what the roofline's HIP events mean); the
  # unprofiled line below is the default command: two batches in flight on the ResNets
  BENCH="python3 $ROOT/bench.py --arch $arch --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --tiles $TILES --interleave 1"
  # untimed: the tile table is pinned so that the tuner's trial launches stay out of the statistics (tools/pick_tiles.py: the
  # fastest of several tuner runs on the whole forward)
  [ -f $TILES ] || timeout 300 python3 $ROOT/tools/pick_tiles.py --arch $arch --dtype $dt --out $TILES > $D/pick_tiles.log 2>&1
  cp $TILES $D/tiles_bs128.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- $BENCH > $D/bench_trace.log 2>&1 || echo "$cfg trace failed"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $BENCH --no-kernel-events > $D/bench_fetch.log 2>&1 || echo "$cfg fetch failed"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $BENCH --no-kernel-events > $D/bench_write.log 2>&1 || echo "$cfg write failed"
  if [ "$cfg" = "dconv:f32" ]; then
    timeout 300 python3 $ROOT/bench.py --steps 20 --warmup 5 --tiles $TILES > $D/bench_unprofiled.json 2> $D/bench_unprofiled.err
  else
    timeout 300 python3 $ROOT/bench.py --arch $arch --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --tiles $TILES > $D/bench_unprofiled.json 2> $D/bench_unprofiled.err
  fi
  echo "$cfg done: $(grep -o '"value": [0-9.]*' $D/bench_unprofiled.json | head -1)"
done
# ---- train step (config 4's per-GPU batch), both compute types: tile table pinned (no tuner launches in the profiled runs), kernel trace,
# FETCH_SIZE / WRITE_SIZE passes, unprofiled line ----
for dt in bf16 f32; do
  T=$OUT/train_$dt
  mkdir -p $T
  TILES=$ROOT/profiles/${TAG}_train_${dt}_tiles.json
  BENCH="python3 $ROOT/bench.py --mode train --dtype $dt --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --tiles $TILES"
  [ -f $TILES ] || timeout 300 $BENCH --no-kernel-events > $T/pick_tiles.log 2>&1      # first run times the tiles on this GPU and writes the table
  cp $TILES $T/tiles_bs32.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -- $BENCH --no-kernel-events > $T/bench_trace.log 2>&1 || echo "train $dt trace failed"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/fetch -- $BENCH --no-kernel-events > $T/bench_fetch.log 2>&1 || echo "train $dt fetch failed"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/write -- $BENCH --no-kernel-events > $T/bench_write.log 2>&1 || echo "train $dt write failed"
  timeout 300 python3 $ROOT/bench.py --mode train --dtype $dt --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --tiles $TILES > $T/bench_unprofiled.json 2> $T/bench_unprofiled.err
  echo "train $dt done: $(grep -o '"value": [0-9.]*' $T/bench_unprofiled.json | head -1)"
done
# ---- the HBM-bound kernels one by one ----
mkdir -p $OUT/micro
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/micro/trace -- python3 $ROOT/tools/bench_micro.py --out $OUT/micro/table.json --md $OUT/micro/table.md > $OUT/micro/bench.log 2>&1 || echo "micro failed"
# keep only the summaries (the raw per-dispatch traces are large)
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
