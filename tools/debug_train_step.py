import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_oracle, pose_oracle, train_oracle
from simple_pose_amd import synth
from simple_pose_amd.nets import pose_resnet_dconv
from simple_pose_amd.train import PoseTrainer
DEV="cuda:0"
B,H,W=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 7)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m=m.to(DEV).train(); sdt={k: torch.from_numpy(v.copy()) for k, v in sd.items()}
x = synth.input_images(B, 7, h=H, w=W); joints = synth.joints_batch(B, 17, seed=47, w=W // 4, h=H // 4)
t, w = pose_oracle.encode_refine(joints, 2.0, (W // 4, H // 4))
tr = PoseTrainer(m, in_h=H, in_w=W)
loss = tr.forward_backward(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
oloss, og, oheat = train_oracle.forward_backward(sdt, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w))
print('loss', loss.item(), float(oloss))
named=dict(m.named_parameters())
rows=[]
for k,g in og.items():
    a=named[k].grad.cpu().double(); b=g.double()
    d=(a-b).abs()
    rows.append((float(d.max()/(b.pow(2).mean().sqrt()+1e-30)), float((a-b).norm()/(b.norm()+1e-30)), float((d>1e-3*b.abs().max()).float().mean()), k))
rows.sort(reverse=True)
for r in rows[:12]: print("max/rms %.3e  L2rel %.3e  frac>1e-3max %.4f  %s"%r)
print('median L2rel', np.median([r[1] for r in rows]))
