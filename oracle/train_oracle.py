"""oracle/train_oracle.py - TEST INFRASTRUCTURE.  CPU restatement of one training step of the reference
(processors/ddp_pose_resnet_solver.py:110-133): train-mode forward (oracle/nets_oracle.py, torch autograd = the
reference's own third-party arithmetic), loss `0.5 * MSELoss(pred * mask[..., None, None], target * mask[..., None, None])`
(:94,:117), backward, and torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) (:70-72) restated
element-wise from its documented update rule."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import nets_oracle


def forward_backward(sd: Dict[str, torch.Tensor], x: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor,
                     arch: str = "resnet50_dconv", amp_bf16: bool = False) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """Returns (loss, grads by parameter name, heat maps).  `sd` buffers (running stats) are updated in place.
    amp_bf16: run the forward under torch.autocast(bfloat16) - the CPU stand-in for the reference's `optim.amp` branch
    (ddp...:121-127: autocast + GradScaler; bf16 needs no scaler)."""
    leaves = {}
    work = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or "running_" in k:
            work[k] = v
        else:
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp_bf16):
        heat = nets_oracle.FORWARDS[arch](work, x, training=True)
    heat = heat.float()
    m = mask[..., None, None]
    loss = 0.5 * torch.nn.functional.mse_loss(heat * m, targets * m)
    loss.backward()
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] += 1
    return loss.detach(), {k: v.grad.float() for k, v in leaves.items()}, heat.detach()


def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: dict, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam single-tensor rule (amsgrad False, weight_decay 0, maximize False), in place on `params`."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m/" + k, torch.zeros_like(p))
        v = state.setdefault("v/" + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def gradient_sketch_vector(key: str, i: int, shape):
    """The fixed pseudo-random vector (uniform in [-1, 1), keyed by parameter name) that golden G6b's gradient sketches are taken against
    (oracle/gen_golden.py::gen_train_b8): <grad, r> sums over every element, so fp32 summation noise averages out where the max over a
    gradient slice picks its worst element."""
    import numpy as np

    from simple_pose_amd import synth
    return synth.tensor_uniform(7, f"sketch{i}/{key}", tuple(shape), -1.0, 1.0).astype(np.float64)
