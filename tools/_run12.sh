set -u
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2s2_gputests.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputests.log | tail -4
for b in 16 32 64; do
timeout 300 python bench.py --arch dconv --dtype bf16 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r2s2_layers_dconv_bf16_b$b.json > gpurun_out/r2s2_bench_dconv_bf16_b$b.json 2> gpurun_out/r2s2_bench_dconv_bf16_b$b.err; echo "bf16 b$b rc=$?"
done
for b in 16 32; do
timeout 300 python bench.py --arch dconv --batch $b --steps 20 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r2s2_layers_dconv_f32_b$b.json > gpurun_out/r2s2_bench_dconv_f32_b$b.json 2> gpurun_out/r2s2_bench_dconv_f32_b$b.err; echo "f32 b$b rc=$?"
done
timeout 300 python bench.py --arch dconv --steps 10 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r2s2_layers_dconv_f32_b128.json > gpurun_out/r2s2_bench_dconv_f32_b128.json 2> /dev/null; echo "f32 b128 rc=$?"
