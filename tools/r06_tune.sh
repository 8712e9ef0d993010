#!/bin/bash
# Round 6: parity of the four-MFMA-wave ring tiles, then the per-arch tile tuner with the new candidates (tracked r05 tables compete).
# Run on the GPU box: bash tools/r06_tune.sh   (writes gpurun_out/r06_*; copy the winners to profiles/)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ring_kernel_matches or conv_kernel_name" > gpurun_out/r06_ring_parity.log 2>&1 || { tail -30 gpurun_out/r06_ring_parity.log; exit 1; }
tail -2 gpurun_out/r06_ring_parity.log
for arch in dconv duc hrnet_w32; do
  cp profiles/r05_${arch}_bf16_tiles.json gpurun_out/r06_${arch}_bf16_tiles.json
  timeout -k 10 500 python tools/pick_tiles.py --arch $arch --dtype bf16 --tunes 2 --out gpurun_out/r06_${arch}_bf16_tiles.json > gpurun_out/r06_pick_${arch}.log 2>&1 || { tail -20 gpurun_out/r06_pick_${arch}.log; exit 1; }
  tail -12 gpurun_out/r06_pick_${arch}.log
done
