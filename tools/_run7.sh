set -u
timeout 600 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --out gpurun_out/layers_ring3_dconv.json 2>&1 | grep -v amdgpu.ids
