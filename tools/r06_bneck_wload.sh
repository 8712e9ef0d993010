# same-box A/B: bottleneck_c64_w8_kernel's filters read in memory order (shipped) against the fragment-order gather (SP_BNECK_GATHER_W build)
L=simple_pose_amd/lib
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bottleneck" 2>&1 | tail -1
for i in 1 2 3; do
  echo -n "memory order (shipped): "; SIMPLE_POSE_HIP_LIB=$L/libsimple_pose_hip.so python tools/diag_bneck.py 2>&1 | grep "per launch"
  echo -n "fragment-order gather:  "; SIMPLE_POSE_HIP_LIB=$L/bneck_gather.so python tools/diag_bneck.py 2>&1 | grep "per launch"
done
echo -n "bs=32 shipped: "; SIMPLE_POSE_HIP_LIB=$L/libsimple_pose_hip.so python tools/diag_bneck.py --batch 32 2>&1 | grep "per launch"
echo -n "bs=32 gather:  "; SIMPLE_POSE_HIP_LIB=$L/bneck_gather.so python tools/diag_bneck.py --batch 32 2>&1 | grep "per launch"
