# same-box A/B: HRNet-W32 bf16 bs=128 with the 32-channel BasicBlocks as one launch each (SP_HRNET_BLOCKS=1) against one launch per conv (0)
set -e
mkdir -p gpurun_out/r06_blocks
for rep in 1 2; do
for v in 0 1; do
  SP_HRNET_BLOCKS=$v python bench.py --arch hrnet_w32 --dtype bf16 --steps 30 --warmup 8 > gpurun_out/r06_blocks/three_$v.json 2>gpurun_out/r06_blocks/three_$v.err
  SP_HRNET_BLOCKS=$v python bench.py --arch hrnet_w32 --dtype bf16 --steps 30 --warmup 8 --interleave 1 > gpurun_out/r06_blocks/one_$v.json 2>gpurun_out/r06_blocks/one_$v.err
  python - <<PY
import json
a=json.load(open("gpurun_out/r06_blocks/three_$v.json")); b=json.load(open("gpurun_out/r06_blocks/one_$v.json"))
print("SP_HRNET_BLOCKS=$v: default (batches in flight: %s) %.0f img/s; one in flight %.0f img/s" % (a["config"].get("batches_in_flight"), a["value"], b["value"]))
PY
done
done
