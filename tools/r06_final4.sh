#!/bin/bash
# after the fold change: the train-step profiles again (both compute types), the train tests, the default bench command as the driver runs it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
sed -e 's/^timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT\/micro.*$/true/' $ROOT/tools/run_profiles.sh > /tmp/run_profiles_train.sh
bash /tmp/run_profiles_train.sh r06 " " > $ROOT/gpurun_out/run_profiles_r06_train.log 2>&1; tail -4 $ROOT/gpurun_out/run_profiles_r06_train.log
cd $ROOT
s=$(date +%s); timeout 600 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err; echo "default bench rc=$? $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_default_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for o in d["other_configs"]:
    print(" ", str(o.get("config", ""))[:80], o.get("value"), o.get("ms_per_step"), o.get("error"))
PY
