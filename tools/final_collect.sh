#!/bin/bash
# Everything a round's evidence needs, in one GPU call (run through gpurun from the repo root):
#   bash tools/final_collect.sh r02
# GPU tests (measured parity values) -> profiles of every config -> PMC stall tables -> per-layer tables for tools/layer_floors.py
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rA > gpurun_out/final_gputests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/final_gputests.log | tail -2
bash tools/run_profiles.sh $TAG > gpurun_out/run_profiles_$TAG.log 2>&1; tail -6 gpurun_out/run_profiles_$TAG.log
for cfg in "dconv bf16" "duc bf16" "hrnet_w32 bf16" "dconv f32"; do
  set -- $cfg
  bash tools/pmc_bench.sh $1 $2 conv > gpurun_out/pmc_${1}_${2}.md 2> gpurun_out/pmc_${1}_${2}.err
  T=gpurun_out/prof_$TAG/${1}_${2}/tiles_bs128.json
  timeout 300 python bench.py --arch $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --tiles $T --layers-out gpurun_out/final_layers_${1}_${2}.json > /dev/null 2>&1
done
echo collected
