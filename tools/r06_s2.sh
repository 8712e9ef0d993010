#!/bin/bash
# Round 6, GPU session 2: G6b / repack / ring parity, per-layer candidates on the layers the four-MFMA-wave tiles took, repack-occupancy A/B, SQ counters
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu -k "golden_batch_8 or repack_walks or ring_kernel_matches or vs_reference_golden" > gpurun_out/r06_s2_tests.log 2>&1 || { tail -40 gpurun_out/r06_s2_tests.log; exit 1; }
tail -3 gpurun_out/r06_s2_tests.log
timeout -k 10 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only layer3.1.conv2,layer3.1.conv1,layer4.1.conv1,layer4.1.conv2,deconv_layers.0,layer2.1.conv1 --out gpurun_out/r06_layers_lw4_dconv.json > gpurun_out/r06_layers_lw4_dconv.log 2>&1 || { tail -20 gpurun_out/r06_layers_lw4_dconv.log; exit 1; }
cat gpurun_out/r06_layers_lw4_dconv.log | tail -12
bash tools/r05_ab_lib.sh simple_pose_amd/lib/ab_prev_repack.so repack 3 -- --mode train --dtype bf16 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06_ab_repack.log 2>&1 || { tail -20 gpurun_out/r06_ab_repack.log; exit 1; }
cat gpurun_out/r06_ab_repack.log
python - <<'PY'
import glob, json
for tag in ("prev", "new"):
    print(tag, [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"gpurun_out/r05_ab_repack/{tag}_*.json"))])
PY
bash tools/pmc_bench.sh dconv bf16 conv_ring > gpurun_out/r06_pmc_dconv_ring.md 2>&1 || { tail -20 gpurun_out/r06_pmc_dconv_ring.md; exit 1; }
cat gpurun_out/r06_pmc_dconv_ring.md
