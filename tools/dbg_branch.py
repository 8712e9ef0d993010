import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
from test_gpu_train import _model, _batch, DEV
from simple_pose_amd.train import PoseTrainer

B, H, W = 4, 64, 64
x, t, w = _batch(B, H, W, 11)
xd, td, wd = (torch.from_numpy(a).to(DEV) for a in (x, t, w))

def run_steps(mode, n=3, dtype="fp32"):
    os.environ["SP_BRANCH"] = mode
    m, _ = _model(11)
    tr = PoseTrainer(m, in_h=H, in_w=W, lr=1e-3, dtype=dtype)
    for _ in range(n):
        tr.step(xd, td, wd)
    torch.cuda.synchronize()
    return tr.flat.data.clone()

def run_autograd(mode, n=3, dtype="fp32"):
    os.environ["SP_BRANCH"] = mode
    m, _ = _model(11)
    if dtype == "bf16":
        m.compute_dtype = "bf16"
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    for _ in range(n):
        opt.zero_grad()
        p = m(xd)
        loss = 0.5 * crit(p * wd[..., None, None], td * wd[..., None, None])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([q.detach().reshape(-1) for q in m.parameters()]).clone()

for name, fn in (("step", run_steps), ("autograd", run_autograd)):
    ref = fn("0")
    for mode in ("0", "1", "1", "bwd", "fwd"):
        got = fn(mode)
        d = (got - ref).abs()
        print(name, mode, "equal", torch.equal(got, ref), "max dev %.3e" % float(d.max()), "frac>1e-6 %.4f" % float((d > 1e-6).float().mean()), flush=True)
