set -u
mkdir -p gpurun_out
timeout 600 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --out gpurun_out/layers_ring2_dconv.json > gpurun_out/r2_layers2_dconv.log 2>&1; echo "layers rc=$?"
grep -v "amdgpu.ids" gpurun_out/r2_layers2_dconv.log | tail -40
