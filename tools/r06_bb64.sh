set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_basic_block_c32" 2>&1 | tail -2
python tools/diag_bb32.py --channels 64 --h 32 --w 24
python tools/diag_bb32.py --channels 64 --h 32 --w 24
python tools/diag_bb32.py --channels 64 --h 32 --w 24 --batch 32
python tools/diag_bb32.py --channels 64 --h 32 --w 24 --batch 256
python tools/diag_bb32.py --channels 64 --h 48 --w 36
