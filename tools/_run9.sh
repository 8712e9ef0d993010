set -u
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring or bench_self or every_tile or full_size" 2>&1 | tail -5
