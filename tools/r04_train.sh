# round-4 helper (run through gpurun): train-step tests + bench lines + a one-step timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r04_$TAG
mkdir -p $OUT
cd $ROOT
if [ "${2:-tests}" = "tests" ]; then
  timeout 900 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_backward_kernels.py -q -m gpu > $OUT/pytest.log 2>&1 || { grep -E "^(FAILED|ERROR|E  )" $OUT/pytest.log | cut -c1-300 | tail -40; tail -3 $OUT/pytest.log; }
  tail -3 $OUT/pytest.log
fi
cd /tmp && export TMPDIR=/tmp
for dt in bf16 f32; do
  T=$ROOT/profiles/r03_train_${dt}_tiles.json
  timeout 300 python3 $ROOT/bench.py --mode train --dtype $dt --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --tiles $T --layers-out $OUT/layers_$dt.json > $OUT/train_$dt.json 2> $OUT/train_$dt.err || { tail -5 $OUT/train_$dt.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$OUT/train_$dt.json')); print('$dt', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['step_split_ms'])"
done
T=$ROOT/profiles/r03_train_bf16_tiles.json
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --dtype bf16 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --tiles $T --no-kernel-events > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/step_timeline.py $F --list > $OUT/timeline.txt 2>&1
head -24 $OUT/timeline.txt
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
