set -u
mkdir -p gpurun_out
for a in dconv duc hrnet_w32; do
timeout 600 python bench.py --arch $a --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/r2_layers_e2e_${a}_bf16.json --tiles gpurun_out/r2_tiles_${a}_bf16.json > gpurun_out/r2_bench_${a}_bf16.json 2> gpurun_out/r2_bench_${a}_bf16.err; echo "$a rc=$?"; python - <<PY
import json
l=json.load(open("gpurun_out/r2_bench_${a}_bf16.json"))
print(l["value"], l["ms_per_step"], l["roofline"]["kernel"], l["roofline"]["frac"], l["roofline"]["all_conv_kernels"])
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_dconv_f32.json 2> gpurun_out/r2_bench_dconv_f32.err; echo "f32 rc=$?"; python -c "
import json; l=json.load(open('gpurun_out/r2_bench_dconv_f32.json')); print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac'], l['roofline']['traffic'], l['roofline']['all_conv_kernels'])"
