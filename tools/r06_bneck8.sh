#!/bin/bash
# Round 6: bottleneck_c64_w8_kernel (eight waves, two per SIMD; one x fragment load per halo row block and k step) against the four-wave kernel:
# bitwise tests with either, isolated launch, whole forward (same box, alternating)
set -o pipefail
mkdir -p gpurun_out
for w in 1 0; do
  SP_BNECK_W8=$w python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bottleneck" > gpurun_out/r06_bneck8_tests_$w.log 2>&1 || { tail -30 gpurun_out/r06_bneck8_tests_$w.log; exit 1; }
  tail -1 gpurun_out/r06_bneck8_tests_$w.log
done
for i in 1 2 3; do
  for w in 0 1; do echo -n "w8=$w: "; SP_BNECK_W8=$w python tools/diag_bneck.py 2>&1 | grep "per launch"; done
done | tee gpurun_out/r06_bneck8_ab.txt
for i in 1 2 3; do
  for arch in duc dconv; do
    for w in 0 1; do
      echo -n "$arch w8=$w: "; SP_BNECK_W8=$w python bench.py --arch $arch --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
    done
  done
done | tee -a gpurun_out/r06_bneck8_ab.txt
