set -u
mkdir -p gpurun_out
timeout 1150 python -m pytest tests -m gpu -q -rA > gpurun_out/r2s2_gputests.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputests.log | grep -E "passed|failed|^FAILED|^ERROR" | tail -12
timeout 300 python bench.py --mode train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s2_bench_train_bf16.json 2> gpurun_out/r2s2_bench_train_bf16.err; echo "train rc=$?"
cat gpurun_out/r2s2_bench_train_bf16.json
