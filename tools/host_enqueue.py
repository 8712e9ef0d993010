#!/usr/bin/env python3
"""How long the HOST needs to enqueue one step (Program.run + decode) vs how long the GPU needs to execute it (tools, not product):
if the two are close, the step is launch-bound and kernel work will not show.   python tools/host_enqueue.py --arch hrnet_w32 --dtype bf16"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from simple_pose_amd import synth
from simple_pose_amd.metrics import GaussTaylorKeyPointDecoder
from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="hrnet_w32"); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.arch == "hrnet_w32":
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    model = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(model.cfg, 17), seed=0)
else:
    model = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[a.arch].resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict([(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()], seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model = model.to(dev).eval()
if a.dtype == "bf16":
    model.compute_dtype = "bf16"
B = a.batch
x = torch.from_numpy(np.concatenate([synth.input_images(8, 100)] * (B // 8), 0)).to(dev)
tinv = torch.from_numpy(synth.trans_inv_batch(B)).to(dev)
prog = model.hip_program(x)
prog.autotune(x)
dec = GaussTaylorKeyPointDecoder()
for ms in (True, False):
    prog.multi_stream = ms
    for _ in range(5):
        dec(prog.run(x), tinv)
    torch.cuda.synchronize()
    N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        dec(prog.run(x), tinv)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n_ops = len(prog.ops)
    print(f"{a.arch} {a.dtype} multi_stream={ms}: {n_ops} launches/step; host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, "
          f"GPU done after {1e3 * (t2 - t0) / N:.2f} ms/step")
