set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring or every_tile or bench_self or packing or ctypes_only" > gpurun_out/r2s2_gpuring.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gpuring.log | tail -4
timeout 400 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only layer3.1,layer4.1,layer3.0,layer4.0,layer2.1 2>&1 | grep -v amdgpu.ids > gpurun_out/r2s2_layers192.log; tail -60 gpurun_out/r2s2_layers192.log
for a in dconv duc; do
timeout 300 python bench.py --arch $a --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --retune --layers-out gpurun_out/r2s2_layers_${a}_bf16_v2.json > gpurun_out/r2s2_bench_${a}_bf16_v2.json 2> gpurun_out/r2s2_bench_${a}_bf16_v2.err; echo "$a rc=$?"
python -c "
import json; l=json.loads(open('gpurun_out/r2s2_bench_${a}_bf16_v2.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac'])"
done
