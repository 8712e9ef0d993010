#!/bin/bash
mkdir -p gpurun_out/hrt
python tools/r06_hrnet_tiles_try.py gpurun_out/hrt 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  for v in tracked b4_64x128 b4_96x128 b34_96x128 b4_64_b3_96; do
    T=gpurun_out/hrt/hrnet_tiles_$v.json; [ $v = tracked ] && T=profiles/r06_hrnet_w32_bf16_tiles.json
    echo -n "$v: "; python bench.py --arch hrnet_w32 --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events --tiles $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
  done
done
