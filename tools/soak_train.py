"""Soak: N full-size bf16 train steps (32 x 256 x 192) twice with the projection shortcuts on the branch stream and once without; the three
runs must end on the same bits (same kernels, same accumulation order: a difference is a race)."""
import os, sys, glob, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simple_pose_amd import synth
from simple_pose_amd.nets import pose_resnet_dconv
from simple_pose_amd.commons.transforms import RefineSimpleTransform
from simple_pose_amd.train import PoseTrainer
from oracle import nets_oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev, B = torch.device("cuda", 0), 32
x = torch.from_numpy(np.concatenate([synth.input_images(8, seed=100)] * 4, 0)[:B]).to(dev)
joints = torch.from_numpy(synth.joints_batch(B, 17, seed=200)).to(dev)
targets, mask = RefineSimpleTransform.get_heat_map(joints, 2.0, (48, 64))
tiles = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_bf16_tiles.json")))[-1]))


def run(branch, lazy=True):
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), seed=0).items()}
    model = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17); model.load_state_dict(sd, strict=True); model.to(dev).train()
    tr = PoseTrainer(model, lr=1e-3, dtype="bf16")
    tr.overlap_shortcut, tr.lazy_residual_grad = branch, lazy
    tr.set_tiles(tiles, B)
    for i in range(N):
        loss = tr.step(x, targets, mask)
    torch.cuda.synchronize()
    return float(loss), tr.flat.data.clone(), torch.cat([b.reshape(-1).float() for b in model.buffers()])


ref = run(False, False)
for cfg in ((True, True), (True, True), (False, True), (True, False)):
    got = run(*cfg)
    print("branch, lazy =", cfg, "loss", got[0], "params equal", torch.equal(got[1], ref[1]), "buffers equal", torch.equal(got[2], ref[2]), flush=True)
print("reference loss", ref[0])
