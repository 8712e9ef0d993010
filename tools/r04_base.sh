set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_base
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$ROOT/profiles/r03_train_bf16_tiles.json
timeout 300 python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --tiles $T > $OUT/train_bf16.json 2> $OUT/train_bf16.err
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --tiles $T --no-kernel-events > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/step_timeline.py $F --list > $OUT/timeline.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
tail -3 $OUT/train_bf16.json | cut -c1-400
