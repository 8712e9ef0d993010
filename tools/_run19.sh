set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "basic_block or hrnet" > gpurun_out/r2s2_gpubb.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gpubb.log | tail -4
timeout 300 python bench.py --arch hrnet_w32 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --retune --layers-out gpurun_out/r2s2_layers_hrnet_bf16_v4.json > gpurun_out/r2s2_bench_hrnet_w32_bf16_v4.json 2> gpurun_out/r2s2_bench_hrnet_w32_bf16_v4.err; echo "hrnet rc=$?"
python -c "
import json; l=json.loads(open('gpurun_out/r2s2_bench_hrnet_w32_bf16_v4.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac']); print({k[:40]:(v['launches'],v['avg_us']) for k,v in l['roofline']['by_kernel'].items()})"
