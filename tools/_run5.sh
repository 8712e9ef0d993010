set -u
ONLY=deconv_layers,layer3.1.conv2,layer4.1.conv2,layer2.1.conv2,layer3.1.conv1,layer3.0.conv3
echo "== v1"
SIMPLE_POSE_HIP_LIB=$PWD/tools/_tmp/libv1.so timeout 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only $ONLY 2>&1 | grep -v amdgpu.ids
echo "== v2 SPREAD 2"
SP_RING_SPREAD=2 timeout 300 python tools/bench_conv_layers.py --arch dconv --dtype bf16 --only $ONLY 2>&1 | grep -v amdgpu.ids
