set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_basic_block or fused_basic_blocks" 2>&1 | tail -1
python tools/diag_bb32.py
python tools/diag_bb32.py --stamps
python tools/diag_bb32.py --batch 32
python tools/diag_bb32.py --batch 64
