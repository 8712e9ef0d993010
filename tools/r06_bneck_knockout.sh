#!/bin/bash
# Diagnostic (round 6): which stream bounds bottleneck_c64_w8_kernel? Time the launch with one stream knocked out at a time (results are wrong by construction).
L=simple_pose_amd/lib
for i in 1 2; do
  echo -n "full:            "; SIMPLE_POSE_HIP_LIB=$L/libsimple_pose_hip.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "no y stores:     "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_1.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "no residual:     "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_2.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "no x loads:      "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_3.so python tools/diag_bneck.py 2>&1 | grep "per launch"
done | tee gpurun_out/r06_bneck_knockout.txt
