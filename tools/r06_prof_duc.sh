#!/bin/bash
# re-profile DUC bf16 after the tile table learned the head kernel (the other configs' profiles are unchanged)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
sed -e 's/^for dt in bf16 f32; do/for dt in ; do/' -e 's/^timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT\/micro.*$/true/' $ROOT/tools/run_profiles.sh > /tmp/run_profiles_duc.sh
bash /tmp/run_profiles_duc.sh r06 "duc:bf16" > $ROOT/gpurun_out/run_profiles_r06_duc.log 2>&1; tail -4 $ROOT/gpurun_out/run_profiles_r06_duc.log
