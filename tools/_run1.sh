set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputests1.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r2_gputests1.log
timeout 600 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --out gpurun_out/layers_ring_dconv.json > gpurun_out/r2_layers_dconv.log 2>&1; echo "layers rc=$?"
tail -70 gpurun_out/r2_layers_dconv.log
