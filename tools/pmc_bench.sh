#!/bin/bash
# SQ stall / LDS / instruction-mix counters per kernel of one bench.py configuration (two --pmc passes, no trace domains):
#   bash tools/pmc_bench.sh hrnet_w32 bf16 [kernel-substring]
ARCH=${1:-dconv}; DT=${2:-bf16}; KSUB=${3:-kernel}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bench_${ARCH}_${DT}
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --arch $ARCH --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --single-stream"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
OUT=$OUT KSUB=$KSUB python3 - <<'PY'
import csv, glob, collections, os
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for tag in "ab":
    for f in glob.glob(os.environ["OUT"] + "/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if os.environ["KSUB"] not in k: continue
            k=k.replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:70]
            d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
print("| kernel | launches | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | WAIT_INST_LDS | LDS conflict / LDS active | MFMA busy share | wave-cycles / launch |")
print("|---|---|---|---|---|---|---|---|---|")
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["SQ_WAVE_CYCLES"]):
    w=v["SQ_WAVE_CYCLES"] or 1
    L=n[(k,"SQ_WAVE_CYCLES")]
    mf=v["SQ_VALU_MFMA_BUSY_CYCLES"]/max(v["GRBM_GUI_ACTIVE"],1)/128.0
    print(f"| `{k}` | {L} | {v['SQ_WAIT_ANY']/w:.3f} | {v['SQ_WAIT_INST_ANY']/w:.3f} | {v['SQ_ACTIVE_INST_ANY']/w:.3f} | {v['SQ_WAIT_INST_LDS']/w:.3f} | {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f} | {mf:.3f} | {w/max(L,1):.3g} |")
PY
