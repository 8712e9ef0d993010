#!/bin/bash
# re-profile ONE config after a kernel change (edit the config list below; the other configs' profiles stay), then the default bench command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
sed -e 's/^for dt in bf16 f32; do/for dt in ; do/' -e 's/^timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT\/micro.*$/true/' $ROOT/tools/run_profiles.sh > /tmp/run_profiles_bf16.sh
bash /tmp/run_profiles_bf16.sh r06 "hrnet_w32:bf16" > $ROOT/gpurun_out/run_profiles_r06_bf16.log 2>&1; tail -4 $ROOT/gpurun_out/run_profiles_r06_bf16.log
cd $ROOT
s=$(date +%s); timeout 600 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err; echo "default bench rc=$? $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_default_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for o in d["other_configs"]:
    print(" ", o.get("config", "")[:60], o.get("value"), o.get("ms_per_step"), o.get("error"))
PY
