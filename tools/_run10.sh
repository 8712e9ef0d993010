set -u
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "autograd or needs_no_trainer" 2>&1 | grep -v "amdgpu.ids" | tail -40
