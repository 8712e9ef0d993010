#!/usr/bin/env python3
"""Round 6 experiment: HRNet-W32 bf16 tile tables with the four-MFMA-wave ring tiles forced onto the low-resolution branches (branch 3: 128 channels at 16x12,
M = 24,576; branch 4: 256 channels at 8x6, M = 6,144 = 96 tiles of 128x128 on 256 CUs).  Writes candidate tables next to the tracked one; bench.py --tiles times them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simple_pose_amd import _lib, synth
from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes

out = sys.argv[1]
model = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(model.cfg, 17), seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model = model.to("cuda:0").eval()
model.compute_dtype = "bf16"
x = torch.from_numpy(synth.input_images(8, 100)).to("cuda:0")
prog = model.hip_program(x)
base = json.load(open(os.path.join(ROOT, "profiles", "r06_hrnet_w32_bf16_tiles.json")))
lib = _lib.lib()
variants = {"b4_64x128": {}, "b4_96x128": {}, "b34_96x128": {}, "b4_64_b3_96": {}}
for op in prog.ops:
    if op.kind != "conv" or op.name not in base:
        continue
    d = op.desc
    d.batch = 128
    M = 128 * d.grid_h * d.grid_w
    if tuple(base[op.name])[:2] != (128, 128) or len(base[op.name]) < 3 or base[op.name][2] not in (1, 3):
        continue
    def ok(bm, bn):
        keep = (d.tile_m, d.tile_n, d.kernel)
        d.tile_m, d.tile_n, d.kernel = bm, bn, 4
        r = lib.sp_conv2d_ring_ok(d)
        d.tile_m, d.tile_n, d.kernel = keep
        return r
    if M == 6144:
        if ok(64, 128): variants["b4_64x128"][op.name] = [64, 128, 4]; variants["b4_64_b3_96"][op.name] = [64, 128, 4]
        if ok(96, 128): variants["b4_96x128"][op.name] = [96, 128, 4]; variants["b34_96x128"][op.name] = [96, 128, 4]
    elif M == 24576:
        if ok(96, 128): variants["b34_96x128"][op.name] = [96, 128, 4]; variants["b4_64_b3_96"][op.name] = [96, 128, 4]
for name, ch in variants.items():
    t = dict(base)
    t.update(ch)
    json.dump(t, open(os.path.join(out, f"hrnet_tiles_{name}.json"), "w"))
    print(name, len(ch), "layers changed")
