#!/bin/bash
# Round 6, after the fused BasicBlocks became HRNet's default: full -m gpu suite, HRNet-W32's rocprofv3 profiles again, the default bench command as the driver runs it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -x -q -rA > gpurun_out/final2_gputests.log 2>&1; echo "pytest rc=$?"
grep -E " passed| failed" gpurun_out/final2_gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
sed -e 's/^for dt in bf16 f32; do/for dt in ; do/' -e 's/^timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT\/micro.*$/true/' $ROOT/tools/run_profiles.sh > /tmp/run_profiles_hrnet.sh
rm -rf gpurun_out/prof_r06/hrnet_w32_bf16
bash /tmp/run_profiles_hrnet.sh r06 "hrnet_w32:bf16" > $ROOT/gpurun_out/run_profiles_r06_hrnet.log 2>&1; tail -3 $ROOT/gpurun_out/run_profiles_r06_hrnet.log
cd $ROOT
s=$(date +%s); timeout 600 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err; echo "default bench rc=$? $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_default_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for o in d["other_configs"]:
    print(" ", str(o.get("config", ""))[:80], o.get("value"), o.get("ms_per_step"), o.get("error"))
PY
