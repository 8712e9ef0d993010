# same-box A/B: HRNet-W32 bf16 with the 64-channel BasicBlocks as one launch each (SP_HRNET_BLOCKS64=1) against one launch per conv (0); the 32-channel blocks fused in both
set -e
for rep in 1 2; do
for v in 0 1; do
  a=$(SP_HRNET_BLOCKS64=$v python bench.py --arch hrnet_w32 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(SP_HRNET_BLOCKS64=$v python bench.py --arch hrnet_w32 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events --interleave 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  c=$(SP_HRNET_BLOCKS64=$v python bench.py --arch hrnet_w32 --dtype bf16 --batch 32 --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "SP_HRNET_BLOCKS64=$v: bs=128 three in flight $a img/s; one in flight $b; bs=32 default $c"
done
done
