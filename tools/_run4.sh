set -u
for sp in 2 4; do
  echo "== SPREAD $sp"
  SP_RING_SPREAD=$sp timeout 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only deconv_layers,layer3.1.conv2,layer4.1.conv2,layer2.1.conv2,layer3.1.conv1,layer3.0.conv3 2>&1 | grep -v amdgpu.ids
done
