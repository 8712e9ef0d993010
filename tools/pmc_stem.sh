#!/bin/bash
# Counters of the fused stem launches (tools/diag_stem.py with the SHIPPED library: bs=128, 256x192, fp32 and bf16): separate --pmc passes, no
# trace domains.   bash tools/pmc_stem.sh > gpurun_out/pmc_stem.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_stem
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -f $ROOT/simple_pose_amd/lib/libsimple_pose_hip_stemdiag.so
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/diag_stem.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA --output-format csv -d $OUT/b -- python3 $ROOT/tools/diag_stem.py > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $ROOT/tools/diag_stem.py > $OUT/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $ROOT/tools/diag_stem.py > $OUT/w.log 2>&1
grep "us per launch" $OUT/a.log
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for tag in "abfw":
    for f in glob.glob(os.environ["OUT"] + "/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "stem_pool_kernel" not in r["Kernel_Name"]: continue
            k = "bf16" if "<true" in r["Kernel_Name"] else "fp32"
            d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
names = sorted(set(d["fp32"]) | set(d["bf16"]))
print("| counter | fp32 per launch | bf16 per launch |\n|---|---|---|")
for c in names:
    print("| %s | %s | %s |" % (c, *(("%.0f" % (d[k][c]/n[k][c])) if n[k][c] else "-" for k in ("fp32", "bf16"))))
for k in ("fp32", "bf16"):
    if n[k]["FETCH_SIZE"] and n[k]["WRITE_SIZE"]:
        print(f"\n{k}: HBM bytes per launch (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, KiB units): {(2*d[k]['FETCH_SIZE']/n[k]['FETCH_SIZE'] + d[k]['WRITE_SIZE']/n[k]['WRITE_SIZE'])*1024/1e6:.1f} MB")
    if n[k]["SQ_VALU_MFMA_BUSY_CYCLES"] and n[k]["SQ_BUSY_CYCLES"]:
        print(f"{k}: MFMA busy / SQ busy cycles = {d[k]['SQ_VALU_MFMA_BUSY_CYCLES']/n[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (d[k]['SQ_BUSY_CYCLES']/n[k]['SQ_BUSY_CYCLES']):.3f} (both summed over the chip's SQs)")
PY
