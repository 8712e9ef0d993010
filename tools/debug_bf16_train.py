import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_oracle, pose_oracle, train_oracle
from simple_pose_amd import synth
from simple_pose_amd.nets import pose_resnet_dconv
from simple_pose_amd.train import PoseTrainer
DEV="cuda:0"
B,H,W=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
def mk():
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 9)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(DEV).train(), {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
x = synth.input_images(B, 9, h=H, w=W); joints = synth.joints_batch(B, 17, seed=49, w=W // 4, h=H // 4)
t, w = pose_oracle.encode_refine(joints, 2.0, (W // 4, H // 4))
xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
m16,_=mk(); tr16 = PoseTrainer(m16, in_h=H, in_w=W, dtype="bf16"); l16=tr16.forward_backward(xs,ts,ws).item()
m32,sd=mk(); tr32 = PoseTrainer(m32, in_h=H, in_w=W); l32=tr32.forward_backward(xs,ts,ws).item()
print('loss bf16', l16, 'fp32', l32)
n16=dict(m16.named_parameters()); n32=dict(m32.named_parameters())
rows=[(float((n16[k].grad-n32[k].grad).norm()/(n32[k].grad.norm()+1e-30)),k) for k in n32]
for k in ['final_layer.weight','final_layer.bias','deconv_layers.7.weight','deconv_layers.6.weight','deconv_layers.4.weight','deconv_layers.3.weight','deconv_layers.1.weight','deconv_layers.0.weight','layer4.2.conv3.weight','layer4.0.conv1.weight','layer3.5.conv3.weight','layer3.0.conv1.weight','layer2.0.conv1.weight','layer1.0.conv1.weight','conv1.weight']:
    print(f"{k:28s} L2rel bf16 vs fp32(HIP) {dict((b,a) for a,b in rows)[k]:.3f}")
