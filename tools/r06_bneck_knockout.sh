#!/bin/bash
# Diagnostic (round 6): which stream bounds bottleneck_c64_w8_kernel? Time the launch with one part knocked out at a time (results are wrong by construction;
# the libraries are SP_BNECK_KNOCKOUT=<k> builds of conv_bneck.hip, never the shipped one).
L=simple_pose_amd/lib
for i in 1 2; do
  echo -n "full:                      "; SIMPLE_POSE_HIP_LIB=$L/libsimple_pose_hip.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "1 no y stores:             "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_1.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "3 no x loads:              "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_3.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "4 stage B: 1 MFMA in 4:    "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_4.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "5 stage C: no transposes:  "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_5.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "6 no stage A:              "; SIMPLE_POSE_HIP_LIB=$L/bneck_knockout_6.so python tools/diag_bneck.py 2>&1 | grep "per launch"
done | tee gpurun_out/r06_bneck_knockout.txt
