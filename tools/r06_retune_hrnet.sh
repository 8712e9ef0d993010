# after the fused BasicBlocks: does another tile table beat the tracked one on HRNet-W32 bf16?  (pick_tiles times one forward in flight; the A/B below is the benchmark's mode)
mkdir -p gpurun_out
cp profiles/r06_hrnet_w32_bf16_tiles.json gpurun_out/r06b_hrnet_w32_bf16_tiles.json
timeout -k 10 500 python tools/pick_tiles.py --arch hrnet_w32 --dtype bf16 --tunes 3 --out gpurun_out/r06b_hrnet_w32_bf16_tiles.json > gpurun_out/r06b_pick_hrnet.log 2>&1; grep -v amdgpu.ids gpurun_out/r06b_pick_hrnet.log | tail -9
for i in 1 2 3; do
  for T in profiles/r06_hrnet_w32_bf16_tiles.json gpurun_out/r06b_hrnet_w32_bf16_tiles.json; do
    echo -n "$T: "; python bench.py --arch hrnet_w32 --dtype bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --tiles $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['one_batch_in_flight']['value'])"
  done
done
