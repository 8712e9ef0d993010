# HRNet-W32 bf16 bs=128 with the fused blocks: how many forwards in flight / per-branch streams
for rep in 1 2; do
for il in 2 3 4; do
  echo -n "interleave $il: "; python bench.py --arch hrnet_w32 --dtype bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --interleave $il 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo -n "interleave 2 + per-branch streams: "; python bench.py --arch hrnet_w32 --dtype bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --interleave 2 --multi-stream 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
