for i in 1 2 3; do
  echo -n "duc igemm final: "; python bench.py --arch duc --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events --tiles simple_pose_amd/lib/duc_prev_tiles.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
  echo -n "duc head kernel: "; python bench.py --arch duc --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
done
