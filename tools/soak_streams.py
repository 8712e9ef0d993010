"""Soak check of the multi-stream schedules (run on the GPU box): HRNet lanes vs one stream over 300 runs, and the streamed train step
(wgrad stream + optimizer in backward) vs the plain schedule over 12 steps - everything must stay bit-identical."""
import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from simple_pose_amd import synth
from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
from simple_pose_amd.nets import pose_resnet_dconv
from simple_pose_amd.train import PoseTrainer
from simple_pose_amd.commons.transforms import RefineSimpleTransform
dev = "cuda:0"
root = os.getcwd()
# 1. HRNet lanes: 300 runs, alternating batch sizes and dtypes, all equal to the single-stream result
net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.to(dev).eval()
bad = 0
for dt in ("fp32", "bf16"):
    net.compute_dtype = dt
    for B in (3, 32):
        x = torch.from_numpy(synth.input_images(B, 5)).to(dev)
        prog = net.hip_program(x)
        prog.multi_stream = False; ref = prog.run(x).clone(); prog.multi_stream = True
        for it in range(75):
            y = prog.run(x)
            if it % 5 == 0: torch.cuda.synchronize()
            bad += int(not torch.equal(y, ref))
print("hrnet lane mismatches", bad)
# 2. trainer: streamed vs plain, 12 steps, bs 8 at 256x192, bf16 and fp32
for dt in ("bf16", "fp32"):
    outs = []
    for streamed in (True, False):
        m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
        layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(layout, seed=2).items()})
        m = m.to(dev).train()
        tr = PoseTrainer(m, lr=1e-3, dtype=dt, overlap_wgrad=streamed, bucket_mb=4.0)
        tr.fuse_optimizer = streamed
        x = torch.from_numpy(synth.input_images(8, 7)).to(dev)
        j = torch.from_numpy(synth.joints_batch(8, 17, seed=9)).to(dev)
        t, w = RefineSimpleTransform.get_heat_map(j, 2.0, (48, 64))
        losses = [tr.step(x, t, w).item() if i % 4 == 3 else float(tr.step(x, t, w)[0]) for i in range(12)]
        torch.cuda.synchronize()
        outs.append((losses, tr.flat.data.clone()))
    print(dt, "losses equal", outs[0][0] == outs[1][0], "params equal", torch.equal(outs[0][1], outs[1][1]), outs[0][0][0], outs[0][0][-1])
