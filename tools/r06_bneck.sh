#!/bin/bash
# Round 6: bottleneck_c64_kernel with coalesced x loads: bitwise tests, stage stamps, same-box A/B against the round-5 kernel
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bottleneck" > gpurun_out/r06_bneck_tests.log 2>&1 || { tail -30 gpurun_out/r06_bneck_tests.log; exit 1; }
tail -2 gpurun_out/r06_bneck_tests.log
python tools/diag_bneck.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_bneck_diag.txt
L=simple_pose_amd/lib
for i in 1 2 3; do
  SIMPLE_POSE_HIP_LIB=$L/ab_prev_bneck.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  SIMPLE_POSE_HIP_LIB=$L/libsimple_pose_hip.so python tools/diag_bneck.py 2>&1 | grep "per launch"
done | tee gpurun_out/r06_bneck_ab.txt
for i in 1 2; do
  for lib in ab_prev_bneck.so libsimple_pose_hip.so; do
    echo -n "$lib dconv bf16: "; SIMPLE_POSE_HIP_LIB=$L/$lib python bench.py --arch dconv --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r06_bneck_ab_dconv.txt
