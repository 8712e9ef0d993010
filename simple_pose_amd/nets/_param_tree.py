"""A module tree built from a list of state_dict keys: owns parameters/buffers under exactly those names.

Used for networks whose reference module tree is deep and irregular (HRNet: 1,754 keys): generating the tree from the
key list guarantees `state_dict()` / `load_state_dict(strict=True)` compatibility with reference checkpoints without
re-typing the reference's class hierarchy.  Leaves named running_mean / running_var / num_batches_tracked are buffers,
everything else is an nn.Parameter.
"""
from __future__ import annotations

import torch
from torch import nn

_BUFFERS = ("running_mean", "running_var", "num_batches_tracked")


class ParamTree(nn.Module):
    def __init__(self, shapes):
        super().__init__()
        for key, shape, dtype in shapes:
            *path, leaf = key.split(".")
            mod = self
            for name in path:
                if name not in mod._modules:
                    mod.add_module(name, nn.Module())
                mod = mod._modules[name]
            if leaf in _BUFFERS:
                t = torch.zeros(shape, dtype=torch.int64 if leaf == "num_batches_tracked" else torch.float32)
                if leaf == "running_var":
                    t.fill_(1.0)
                mod.register_buffer(leaf, t)
            else:
                mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape, dtype=torch.float32)))
