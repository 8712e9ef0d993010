set -u
ONLY=deconv_layers,layer3.1.conv2,layer4.1.conv2,layer2.1.conv2,layer3.1.conv1,layer3.0.conv3
for pf in 0 4 8; do
echo "== PF $pf"
SP_RING_PF=$pf timeout 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only $ONLY 2>&1 | grep -v amdgpu.ids
done
