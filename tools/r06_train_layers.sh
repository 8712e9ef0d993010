python bench.py --mode train --dtype bf16 --batch 32 --steps 10 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r06_train_layers.json > gpurun_out/r06_train_line.json 2>/dev/null
timeout -k 10 600 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --batch 32 --out gpurun_out/r06_layers_dconv_bf16_b32.json 2>&1 | grep -v amdgpu.ids | tail -60
