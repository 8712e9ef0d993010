set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r2s2_gputrain.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputrain.log | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_dconv_f32.json 2> gpurun_out/r2s2_bench_dconv_f32.err; echo "f32 rc=$?"
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_dconv_bf16.json 2> gpurun_out/r2s2_bench_dconv_bf16.err; echo "bf16 rc=$?"
timeout 300 python bench.py --gpus 2 --dist-backend gloo --mode train --dtype bf16 --batch 32 --steps 5 --warmup 2 > gpurun_out/r2s2_bench_train2_sync.json 2> gpurun_out/r2s2_bench_train2_sync.err; echo "train2 sync rc=$?"
timeout 300 python bench.py --gpus 2 --dist-backend gloo --mode train --dtype bf16 --batch 32 --steps 5 --warmup 2 --no-sync-bn > gpurun_out/r2s2_bench_train2_nosync.json 2> gpurun_out/r2s2_bench_train2_nosync.err; echo "train2 nosync rc=$?"
python - <<'PY'
import json,glob
for f in ["gpurun_out/r2s2_bench_dconv_f32.json","gpurun_out/r2s2_bench_dconv_bf16.json","gpurun_out/r2s2_bench_train2_sync.json","gpurun_out/r2s2_bench_train2_nosync.json"]:
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline") or {}
        print(f, l["value"], l["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("traffic"), l.get("step_split_ms"), l.get("collectives_per_step"), l["config"].get("tile_table"))
    except Exception as e:
        print(f, "ERR", e)
PY
