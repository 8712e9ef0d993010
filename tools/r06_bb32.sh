set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_basic_block or fused_basic_blocks" 2>&1 | tail -1
python tools/diag_bb32.py
SP_BB32_W8=0 python tools/diag_bb32.py
python tools/diag_bb32.py --batch 32
python tools/diag_bb32.py --batch 64
python tools/diag_bb32.py --batch 256
python tools/diag_bb32.py --batch 128 --h 96 --w 72
