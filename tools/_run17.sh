set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "direct_3x3" > gpurun_out/r2s2_gpudirect.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gpudirect.log | tail -4
timeout 400 python tools/bench_conv_layers.py --arch hrnet_w32 --dtype bf16 --only stage3.0.branches.1.0,stage2.0.branches.1.0,layer1.1.conv2 2>&1 | grep -v amdgpu.ids | tail -8
timeout 400 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only layer1.1.conv2 2>&1 | grep -v amdgpu.ids | tail -4
for a in hrnet_w32 dconv; do
timeout 300 python bench.py --arch $a --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --retune > gpurun_out/r2s2_bench_${a}_bf16_v3.json 2> gpurun_out/r2s2_bench_${a}_bf16_v3.err; echo "$a rc=$?"
python -c "
import json; l=json.loads(open('gpurun_out/r2s2_bench_${a}_bf16_v3.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac']); print({k:(v['launches'],v['avg_us']) for k,v in l['roofline']['by_kernel'].items()})"
done
