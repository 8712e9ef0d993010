set -u
mkdir -p gpurun_out
for sp in 1 2 3 4; do
  echo "== SPREAD $sp"
  SP_RING_SPREAD=$sp timeout 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only deconv_layers,layer3.1.conv2,layer4.1.conv2,layer2.1.conv2,layer3.1.conv1,layer3.0.conv3 2>&1 | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SP_RING_SPREAD=2 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_ring -- python3 $R/tools/bench_conv_layers.py --arch dconv --dtype bf16 --only deconv_layers.6 --reps 3 --rounds 1 > $R/gpurun_out/pmc_ring.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ.get("GRAFT_REPO_ROOT",".")
fs=glob.glob(root+"/gpurun_out/pmc_ring/**/*counter_collection.csv", recursive=True)
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_" not in k: continue
        k=k[k.index("conv_"):k.index(">")+1]
        d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for k,v in d.items():
    w=v["SQ_WAVE_CYCLES"] or 1
    print(k, "launches",n[k], " ".join(f"{c[3:]}={v[c]/w:.3f}" for c in v if c!="SQ_WAVE_CYCLES"), f"wave_cycles/launch={w/max(n[k],1):.3g}")
PY
