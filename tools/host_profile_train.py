"""cProfile of the host side of the 32-image bf16 train step (what the Python between launches costs)."""
import cProfile, pstats, io, os, sys, glob, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simple_pose_amd import synth
from simple_pose_amd.nets import pose_resnet_dconv
from simple_pose_amd.commons.transforms import RefineSimpleTransform
from simple_pose_amd.train import PoseTrainer
from oracle import nets_oracle
dev, B = torch.device("cuda", 0), 32
sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), seed=0).items()}
model = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17); model.load_state_dict(sd, strict=True); model.to(dev).train()
tr = PoseTrainer(model, lr=1e-3, dtype="bf16")
tr.set_tiles(json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_bf16_tiles.json")))[-1])), B)
x = torch.from_numpy(np.concatenate([synth.input_images(8, seed=100)] * 4, 0)[:B]).to(dev)
joints = torch.from_numpy(synth.joints_batch(B, 17, seed=200)).to(dev)
targets, mask = RefineSimpleTransform.get_heat_map(joints, 2.0, (48, 64))
for _ in range(5):
    tr.step(x, targets, mask)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(x, targets, mask)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
