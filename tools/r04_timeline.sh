# one-step timeline of the bf16 train step (run through gpurun): bash tools/r04_timeline.sh TAG ["ENV=.. ENV=.."]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
OUT=$ROOT/gpurun_out/r04_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in ${2:-}; do export $kv; done
T=${TILES:-$ROOT/profiles/r03_train_bf16_tiles.json}
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --tiles $T --no-kernel-events > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/step_timeline.py $F --list > $OUT/timeline.txt 2>&1
head -24 $OUT/timeline.txt
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
