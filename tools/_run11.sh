set -u
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2s2_gputests.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputests.log | tail -8
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r2s2_bench_dconv_f32.json 2> gpurun_out/r2s2_bench_dconv_f32.err; echo "f32 rc=$?"
for a in dconv duc hrnet_w32; do
timeout 400 python bench.py --arch $a --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r2s2_layers_${a}_bf16.json > gpurun_out/r2s2_bench_${a}_bf16.json 2> gpurun_out/r2s2_bench_${a}_bf16.err; echo "$a rc=$?"
done
timeout 300 python bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_train_bf16.json 2> gpurun_out/r2s2_bench_train_bf16.err; echo "train rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2s2_bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline") or {}
        print(f, l["value"], l["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("traffic"), (r.get("all_conv_kernels") or {}).get("frac"), l.get("step_split_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
