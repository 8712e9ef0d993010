# same-box A/B: stand-alone BatchNorm folds on 16-channel slabs (shipped) against 64-channel slabs (the library before the change); then the train tests
L=simple_pose_amd/lib
for i in 1 2 3; do
  for lib in libsimple_pose_hip.so old_folds.so; do
    echo -n "$lib: "; SIMPLE_POSE_HIP_LIB=$L/$lib python bench.py --mode train --dtype bf16 --batch 32 --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward_kernels.py -x -q -m gpu 2>&1 | tail -2
