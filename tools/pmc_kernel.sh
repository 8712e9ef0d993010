#!/bin/bash
# SQ stall / LDS counters of the conv kernels of ONE layer (tools/bench_conv_layers.py --only <layer>): a --pmc pass, no trace domains.
#   bash tools/pmc_kernel.sh dconv layer1.1.conv2 [kernel-substring]
ARCH=${1:-dconv}; ONLY=${2:-layer1.1.conv2}; KSUB=${3:-conv}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_kernel
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/bench_conv_layers.py --arch $ARCH --dtype bf16 --only $ONLY --reps 3 --rounds 1 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $ROOT/tools/bench_conv_layers.py --arch $ARCH --dtype bf16 --only $ONLY --reps 3 --rounds 1 > $OUT/b.log 2>&1
OUT=$OUT KSUB=$KSUB python3 - <<'PY'
import csv, glob, collections, os
for tag in "ab":
    fs=glob.glob(os.environ["OUT"] + "/%s/**/*counter_collection.csv" % tag, recursive=True)
    d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if os.environ["KSUB"] not in k or "conv" not in k: continue
            k=k[k.index("conv"):][:60]
            d[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            n[(k,r["Counter_Name"])]+=1
    for k,v in d.items():
        print(k)
        for c,val in sorted(v.items()): print(f"   {c:32s} {val/max(n[(k,c)],1):14.0f} per launch")
PY
