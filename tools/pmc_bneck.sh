#!/bin/bash
# Counters of the fused Bottleneck launch (tools/diag_bneck.py with the SHIPPED library: one launch shape, bs=128, 64x48): separate --pmc
# passes, no trace domains.   bash tools/pmc_bneck.sh > gpurun_out/pmc_bneck.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bneck
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -f $ROOT/simple_pose_amd/lib/libsimple_pose_hip_bneckdiag.so
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/diag_bneck.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- python3 $ROOT/tools/diag_bneck.py > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $ROOT/tools/diag_bneck.py > $OUT/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $ROOT/tools/diag_bneck.py > $OUT/w.log 2>&1
grep "us per launch" $OUT/a.log
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
d=collections.defaultdict(float); n=collections.Counter()
for tag in "abfw":
    for f in glob.glob(os.environ["OUT"] + "/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bottleneck_c64_kernel" not in r["Kernel_Name"]: continue
            d[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("| counter | per launch |\n|---|---|")
for c in sorted(d): print(f"| {c} | {d[c]/n[c]:.0f} |")
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    print(f"\nHBM bytes per launch (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, KiB units): {(2*d['FETCH_SIZE']/n['FETCH_SIZE'] + d['WRITE_SIZE']/n['WRITE_SIZE'])*1024/1e6:.1f} MB")
PY
