#!/bin/bash
# Round 6: conv3x3_c128_head_kernel (the DUC head's last layer): bitwise test, isolated candidates, whole-forward DUC bf16
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "head_3x3 or direct_3x3 or forward_vs_reference_golden" > gpurun_out/r06_head_tests.log 2>&1 || { tail -40 gpurun_out/r06_head_tests.log; exit 1; }
tail -2 gpurun_out/r06_head_tests.log
timeout -k 10 300 python tools/bench_conv_layers.py --arch duc --dtype bf16 --only final_layer --out gpurun_out/r06_layers_head_duc.json 2>&1 | grep -v amdgpu.ids | tail -4
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_layers_head_duc.json"))[0]
print({k: v["us"] for k, v in r["cands"].items()})
PY
for i in 1 2 3; do
  echo -n "duc bf16: "; python bench.py --arch duc --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
done | tee gpurun_out/r06_head_duc.txt
