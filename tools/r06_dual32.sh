#!/bin/bash
# Round 6: sp_dual_pw_f32 on the headline configuration: bitwise tests, same-box A/B
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dual or fp32_dual_tail or forward_vs_reference_golden or fused_bottlenecks" > gpurun_out/r06_dual32_tests.log 2>&1 || { tail -40 gpurun_out/r06_dual32_tests.log; exit 1; }
tail -2 gpurun_out/r06_dual32_tests.log
for i in 1 2 3; do
  for t in 0 1; do
    echo -n "dconv f32 tail=$t: "; SP_FUSE_TAIL=$t python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
  done
done | tee gpurun_out/r06_dual32_ab.txt
