# same-box A/B of train-step variants: bash tools/r04_ab2.sh TAG "ENV.. -- extra bench args" ...   (each variant: "ENV1=.. ENV2=.. -- --flag")
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/r04_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=${TILES:-$ROOT/profiles/r04_train_bf16_tiles.json}; case "$T" in /*) ;; *) T=$ROOT/$T;; esac
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    envs="${v%% -- *}"; args=""
    case "$v" in *" -- "*) args="${v#* -- }";; esac
    env $envs timeout 300 python3 $ROOT/bench.py --mode train --dtype ${DT:-bf16} --batch 32 --steps 30 --warmup 8 --no-cpu-baseline --tiles $T --no-kernel-events $args > $OUT/v${i}_$rep.json 2> $OUT/v${i}_$rep.err || { tail -5 $OUT/v${i}_$rep.err; }
    python3 -c "import json; d=json.load(open('$OUT/v${i}_$rep.json')); print('[$v]', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
  done
done
