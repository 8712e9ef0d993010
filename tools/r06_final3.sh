#!/bin/bash
# Round 6 final evidence (one GPU call): full -m gpu suite, smoke, rocprofv3 profiles of every config, the default bench command as the driver runs it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -x -q -rA > gpurun_out/final3_gputests.log 2>&1; echo "pytest rc=$?"
grep -E " passed| failed" gpurun_out/final3_gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/run_profiles.sh r06 > gpurun_out/run_profiles_r06.log 2>&1; tail -8 gpurun_out/run_profiles_r06.log
cd $ROOT
s=$(date +%s); timeout 600 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err; echo "default bench rc=$? $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_default_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for o in d["other_configs"]:
    print(" ", str(o.get("config", ""))[:80], o.get("value"), o.get("ms_per_step"), o.get("error"))
PY
