set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r2s2_gputrain.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputrain.log | tail -5
timeout 300 python bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-train-autotune > gpurun_out/r2s2_bench_train_bf16_notune.json 2> gpurun_out/r2s2_bench_train_bf16_notune.err; echo "train notune rc=$?"
timeout 300 python bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_train_bf16.json 2> gpurun_out/r2s2_bench_train_bf16.err; echo "train rc=$?"
timeout 300 python bench.py --mode train --dtype f32 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_train_f32.json 2> gpurun_out/r2s2_bench_train_f32.err; echo "train f32 rc=$?"
timeout 300 python bench.py --gpus 2 --dist-backend gloo --mode train --dtype bf16 --batch 32 --steps 5 --warmup 2 > gpurun_out/r2s2_bench_train2_sync.json 2> gpurun_out/r2s2_bench_train2_sync.err; echo "train2 sync rc=$?"
timeout 300 python bench.py --gpus 2 --dist-backend gloo --mode train --dtype bf16 --batch 32 --steps 5 --warmup 2 --no-sync-bn > gpurun_out/r2s2_bench_train2_nosync.json 2> gpurun_out/r2s2_bench_train2_nosync.err; echo "train2 nosync rc=$?"
python - <<'PY'
import json,glob
for f in ["gpurun_out/r2s2_bench_train_bf16_notune.json","gpurun_out/r2s2_bench_train_bf16.json","gpurun_out/r2s2_bench_train_f32.json","gpurun_out/r2s2_bench_train2_sync.json","gpurun_out/r2s2_bench_train2_nosync.json"]:
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline") or {}
        print(f, l["value"], l["ms_per_step"], r.get("frac"), l.get("step_split_ms"), l.get("collectives_per_step"), {k:v["ms_per_step"] for k,v in (r.get("by_group") or {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
