#!/bin/bash
# Round 6, late: re-run the tile tuner on the bf16 ResNets with the round's new kernels in the program (dual tail, head kernel, eight-wave bottleneck); the tracked r06 tables compete
set -o pipefail
mkdir -p gpurun_out
for arch in dconv duc; do
  cp profiles/r06_${arch}_bf16_tiles.json gpurun_out/r06b_${arch}_bf16_tiles.json
  timeout -k 10 500 python tools/pick_tiles.py --arch $arch --dtype bf16 --tunes 3 --out gpurun_out/r06b_${arch}_bf16_tiles.json > gpurun_out/r06b_pick_${arch}.log 2>&1 || { tail -20 gpurun_out/r06b_pick_${arch}.log; exit 1; }
  tail -9 gpurun_out/r06b_pick_${arch}.log
done
