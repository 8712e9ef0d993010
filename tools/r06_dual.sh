#!/bin/bash
# Round 6: sp_dual_pw_bf16 (layer1.0's conv3 + projection shortcut as one launch): bitwise tests, same-box A/B on the three bf16 configs
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dual_pointwise or fused_bottlenecks or hrnet_forward or forward_vs_reference_golden or bf16" > gpurun_out/r06_dual_tests.log 2>&1 || { tail -40 gpurun_out/r06_dual_tests.log; exit 1; }
tail -2 gpurun_out/r06_dual_tests.log
for i in 1 2; do
  for arch in duc dconv hrnet_w32; do
    for t in 0 1; do
      echo -n "$arch tail=$t: "; SP_FUSE_TAIL=$t python bench.py --arch $arch --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
    done
  done
done | tee gpurun_out/r06_dual_ab.txt
