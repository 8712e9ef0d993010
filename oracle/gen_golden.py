"""oracle/gen_golden.py - TEST INFRASTRUCTURE.  Run ONCE in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Imports the real reference (oracle/ref_import.py), feeds it build-owned deterministic inputs
(simple_pose_amd/synth.py) and freezes its OUTPUTS as small fixtures under tests/golden/.  Only data
is committed - never reference source.  Fixture inventory:

  g1_dconv_fwd.npz   R50-DConv eval forward, B=2: heat maps + per-stage taps (slice, mean, std, absmax)
  g2_duc_fwd.npz     same for R50-DUC
  g4_decode.npz      decoders (GaussTaylor + Basic + heat_map_to_axis) on: Gaussian maps + noise, noise-like
                     maps, edge cases; identity-scale and random trans_inv
  g1s_dconv_se_fwd.npz  ResNet50-DConv + SELayer (reduction=True) eval forward, B=1, + key list
  g3_hrnet_w32_fwd.npz  HRNet-W32 eval forward, B=1, + the reference's state_dict key/shape list
  g7_next.npz        HeatMapAcc values and collate_fn normalisation
  g11_resnet_variants.npz  resnet18-dconv, resnet34-duc, wide_resnet50_2-dconv, resnet18-dconv+SE (2 images; key lists, sub-sampled maps, key points)
  g12_resnext.npz          resnext50_32x4d-dconv, resnext101_32x8d-duc (grouped 3x3, groups = 32; same content as g11)
  g10_fwd_wide.npz   8 / 8 / 4 / 4 distinct images through DConv / DUC / HRNet-W32 / DConv+SE (sub-sampled maps, per-joint sums, key points)
  g6_train_step.npz  one reference training step (B=2): loss, gradient slices, BN running stats, params after Adam
  g6b_train_step_b8.npz  one reference training step at B=8: loss, the gradient NORM and two sketches of every parameter, slices as in g6
  g5_encode.npz      encoders (Refine + Basic) incl. out-of-range / trunc-toward-zero / vis=0 cases
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True

from oracle import ref_import  # noqa: E402
from simple_pose_amd import synth  # noqa: E402
from oracle.train_oracle import gradient_sketch_vector  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 0


def _stage_taps(model, names):
    taps = {}

    def mk(name):
        def hook(_m, _i, out):
            o = out.detach()
            taps[name + "/slice"] = o[0, :8, :4, :4].numpy().copy()
            taps[name + "/mean"] = np.float64(o.double().mean().item())
            taps[name + "/std"] = np.float64(o.double().std().item())
            taps[name + "/absmax"] = np.float64(o.abs().max().item())
        return hook

    handles = [mod.register_forward_hook(mk(n)) for n, mod in names]
    return taps, handles


def gen_forward(ns):
    torch.set_num_threads(8)
    x = torch.from_numpy(synth.input_images(2, SEED))
    for tag, mod, fname in (("dconv", ns.dconv, "g1_dconv_fwd.npz"), ("duc", ns.duc, "g2_duc_fwd.npz")):
        net = mod.resnet50(pretrained=False, num_classes=17)
        synth.load_conditioned(net, SEED)
        net.eval()
        names = [("maxpool", net.maxpool), ("layer1", net.layer1), ("layer2", net.layer2),
                 ("layer3", net.layer3), ("layer4", net.layer4)]
        if tag == "dconv":
            names += [(f"deconv{i}", net.deconv_layers[3 * i + 2]) for i in range(3)]
        else:
            names += [(f"duc{i}", net.duc_layers[i + 1]) for i in range(2)]
        taps, handles = _stage_taps(net, names)
        with torch.no_grad():
            hm = net(x).numpy()
        for h in handles:
            h.remove()
        n_keys = len(net.state_dict())
        np.savez_compressed(os.path.join(GOLD, fname), heat_maps=hm, seed=SEED, batch=2,
                            n_state_keys=n_keys, **taps)
        print(fname, hm.shape, "absmax", np.abs(hm).max(), "std", hm.std(), "state keys", n_keys)
    return hm


def gen_se(ns):
    """g1s_dconv_se_fwd.npz: ResNet50-DConv with reduction=True (SELayer on the first block of each layer), B=1."""
    torch.set_num_threads(8)
    net = ns.dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
    synth.load_conditioned(net, SEED)
    net.eval()
    x = torch.from_numpy(synth.input_images(1, SEED))
    with torch.no_grad():
        hm = net(x).numpy()
    sd = net.state_dict()
    np.savez_compressed(os.path.join(GOLD, "g1s_dconv_se_fwd.npz"), heat_maps=hm, seed=SEED, batch=1, keys=np.array(list(sd.keys())),
                        shapes=np.array([",".join(str(d) for d in v.shape) for v in sd.values()]))
    print("g1s_dconv_se_fwd.npz", hm.shape, "absmax", np.abs(hm).max(), "keys", len(sd))


def gen_next(ns):
    """g7_next.npz: SURVEY 8(f) rows - HeatMapAcc on network maps vs encoder targets, and the collate_fn normalisation."""
    import importlib, types
    pm = ns.pose_metrics
    acc = pm.HeatMapAcc()
    g1 = np.load(os.path.join(GOLD, "g1_dconv_fwd.npz"))["heat_maps"]
    out = {}
    enc = ns.transforms.RefineSimpleTransform.get_heat_map
    for tag, seed in (("a", 51), ("b", 52)):
        joints = synth.joints_batch(2, 17, seed=seed)
        tgt = np.stack([enc(joints[b], 2.0, (48, 64))[0] for b in range(2)])
        # predictions = targets shifted / perturbed so that some joints hit and some miss
        pred = np.roll(tgt, shift=(2 if tag == "a" else 7), axis=3) + 0.05 * g1
        out[f"acc/{tag}/joints"] = joints
        out[f"acc/{tag}/pred"] = pred.astype(np.float32)
        out[f"acc/{tag}/value"] = np.float32(acc(torch.from_numpy(pred.astype(np.float32)), torch.from_numpy(tgt)).item())
    coco = importlib.import_module("datasets.coco")
    imgs = (synth.tensor_uniform(61, "u8img", (2, 32, 24, 3)) * 255.999).astype(np.uint8)
    items = []
    for i in range(2):
        it = types.SimpleNamespace(img=imgs[i], img_path=f"/x/{i:012d}.jpg", heat_map=np.zeros((17, 8, 6), np.float32),
                                   mask=np.ones(17, np.float32), trans_inv=np.eye(2, 3))
        items.append(it)
    inp, _, _, _, _ = coco.MSCOCO.collate_fn(items)
    out["collate/img_u8"] = imgs
    out["collate/input"] = inp.numpy()
    np.savez_compressed(os.path.join(GOLD, "g7_next.npz"), **out)
    print("g7_next.npz acc", out["acc/a/value"], out["acc/b/value"], "collate", inp.shape, inp.dtype)


def gen_hrnet(ns):
    """g3_hrnet_w32_fwd.npz: HRNet-W32 eval forward, B=1, + the reference's state_dict key list (names/shapes)."""
    torch.set_num_threads(8)
    net = ns.hrnet.get_pose_net(ns.hrnet_w32_yaml, pretrained=None, joint_num=17)
    synth.load_conditioned(net, SEED)
    net.eval()
    x = torch.from_numpy(synth.input_images(1, SEED))
    names = [("layer1", net.layer1)]
    taps, handles = _stage_taps(net, names)
    with torch.no_grad():
        hm = net(x).numpy()
    for h in handles:
        h.remove()
    sd = net.state_dict()
    keys = np.array(list(sd.keys()))
    shapes = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    np.savez_compressed(os.path.join(GOLD, "g3_hrnet_w32_fwd.npz"), heat_maps=hm, seed=SEED, batch=1, keys=keys,
                        shapes=shapes, **taps)
    print("g3_hrnet_w32_fwd.npz", hm.shape, "absmax", np.abs(hm).max(), "std", hm.std(), "keys", len(keys))


def gen_train(ns):
    """g6_train_step.npz: ONE training step of the real reference (ddp...:110-119), B=2: loss, gradient slices, BN running
    stats, parameter slices after torch.optim.Adam(lr=1e-3).step()."""
    torch.set_num_threads(8)
    net = ns.dconv.resnet50(pretrained=False, num_classes=17)
    synth.load_conditioned(net, SEED)
    net.train()
    x = torch.from_numpy(synth.input_images(2, SEED))
    joints = synth.joints_batch(2, 17, seed=41)
    enc = ns.transforms.RefineSimpleTransform.get_heat_map
    tw = [enc(joints[b], 2.0, (48, 64)) for b in range(2)]
    targets = torch.from_numpy(np.stack([t for t, _ in tw]))
    mask = torch.from_numpy(np.stack([w for _, w in tw]))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    opt.zero_grad()
    pred = net(x)
    loss = 0.5 * crit(pred.mul(mask[[..., None, None]]), targets.mul(mask[[..., None, None]]))
    loss.backward()
    out = {"loss": np.float32(loss.item()), "joints": joints, "heat_train": pred.detach().numpy()}
    named = dict(net.named_parameters())
    picks = {"conv1.weight": (slice(0, 2),), "layer1.0.conv1.weight": (slice(0, 4),), "layer2.0.conv2.weight": (slice(0, 2), slice(0, 8)),
             "layer2.0.downsample.0.weight": (slice(0, 2), slice(0, 16)), "layer4.2.conv3.weight": (slice(0, 2), slice(0, 8)),
             "deconv_layers.0.weight": (slice(0, 2), slice(0, 2)), "deconv_layers.6.weight": (slice(0, 2), slice(0, 4)),
             "final_layer.weight": (slice(None),), "final_layer.bias": (slice(None),), "bn1.weight": (slice(None),), "bn1.bias": (slice(None),),
             "layer1.0.bn3.weight": (slice(0, 32),), "layer3.2.bn2.bias": (slice(0, 32),), "deconv_layers.4.weight": (slice(0, 32),)}
    for k, sl in picks.items():
        out["grad/" + k] = named[k].grad[sl].numpy().copy()
        out["gradnorm/" + k] = np.float64(named[k].grad.double().norm().item())
    opt.step()
    for k, sl in picks.items():
        out["param/" + k] = named[k].detach()[sl].numpy().copy()
    bufs = dict(net.named_buffers())
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_var", "deconv_layers.7.running_mean"):
        out["buf/" + k] = bufs[k].numpy().copy()
    out["buf/bn1.num_batches_tracked"] = bufs["bn1.num_batches_tracked"].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "g6_train_step.npz"), **out)
    print("g6_train_step.npz loss", out["loss"], "gradnorm conv1", out["gradnorm/conv1.weight"])


def gen_train_b8(ns):
    """g6b_train_step_b8.npz (round 6): ONE training step of the real reference (ddp...:94,110-119) at B = 8, 256x192 - four times G6's samples
    per BatchNorm channel - pinned through quantities the reference's own fp32 arithmetic holds to 1e-3: the fp64 NORM of every parameter's
    gradient (its own 1-thread / 8-thread / fp64 evaluations agree to <= 1.1e-3, median 5e-5; the max over a gradient SLICE moves by 2e-2),
    two sketches <grad, r> per parameter (<= 8.4e-3 of the norm), next to G6's slices, running statistics and parameters after Adam."""
    torch.set_num_threads(8)
    B = 8
    net = ns.dconv.resnet50(pretrained=False, num_classes=17)
    synth.load_conditioned(net, SEED)
    net.train()
    x = torch.from_numpy(synth.input_images(B, SEED))
    joints = synth.joints_batch(B, 17, seed=43)
    enc = ns.transforms.RefineSimpleTransform.get_heat_map
    tw = [enc(joints[b], 2.0, (48, 64)) for b in range(B)]
    targets = torch.from_numpy(np.stack([t for t, _ in tw]))
    mask = torch.from_numpy(np.stack([w for _, w in tw]))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    opt.zero_grad()
    pred = net(x)
    loss = 0.5 * crit(pred.mul(mask[[..., None, None]]), targets.mul(mask[[..., None, None]]))
    loss.backward()
    out = {"loss": np.float32(loss.item()), "joints": joints, "batch": B, "seed": SEED, "heat_train_sub": pred.detach().numpy()[:, :, ::4, ::4].copy()}
    named = dict(net.named_parameters())
    keys = list(named)
    out["keys"] = np.array(keys)
    out["gradnorm"] = np.array([named[k].grad.double().norm().item() for k in keys], dtype=np.float64)
    out["sketch"] = np.array([[float((named[k].grad.double().numpy() * gradient_sketch_vector(k, i, named[k].shape)).sum()) for i in range(2)]
                              for k in keys], dtype=np.float64)
    picks = {"conv1.weight": (slice(0, 2),), "layer1.0.conv1.weight": (slice(0, 4),), "layer2.0.conv2.weight": (slice(0, 2), slice(0, 8)),
             "layer2.0.downsample.0.weight": (slice(0, 2), slice(0, 16)), "layer4.2.conv3.weight": (slice(0, 2), slice(0, 8)),
             "deconv_layers.0.weight": (slice(0, 2), slice(0, 2)), "deconv_layers.6.weight": (slice(0, 2), slice(0, 4)),
             "final_layer.weight": (slice(None),), "final_layer.bias": (slice(None),), "bn1.weight": (slice(None),), "bn1.bias": (slice(None),),
             "layer1.0.bn3.weight": (slice(0, 32),), "layer3.2.bn2.bias": (slice(0, 32),), "deconv_layers.4.weight": (slice(0, 32),)}
    for k, sl in picks.items():
        out["grad/" + k] = named[k].grad[sl].numpy().copy()
    opt.step()
    for k, sl in picks.items():
        out["param/" + k] = named[k].detach()[sl].numpy().copy()
    bufs = dict(net.named_buffers())
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_var", "deconv_layers.7.running_mean"):
        out["buf/" + k] = bufs[k].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "g6b_train_step_b8.npz"), **out)
    print("g6b_train_step_b8.npz loss", out["loss"], "gradnorm conv1", out["gradnorm"][0], "params", len(keys))


def edge_maps():
    """Hand-built [N,64,48] maps exercising decoder branches."""
    H, W = 64, 48
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

    def gauss(cx, cy, amp=1.0):
        return (amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 8.0)).astype(np.float32)

    maps, names = [], []
    maps.append(np.zeros((H, W), np.float32)); names.append("all_zero")
    maps.append(-gauss(20.3, 30.7) - 0.5); names.append("all_negative")
    for cx in (0.0, 1.2, 1.9, 2.1, W - 3.2, W - 2.0, W - 1.0):
        maps.append(gauss(cx, 31.4)); names.append(f"border_x_{cx}")
    for cy in (0.0, 1.3, 2.2, H - 3.1, H - 2.0, H - 1.0):
        maps.append(gauss(22.6, cy)); names.append(f"border_y_{cy}")
    m = np.zeros((H, W), np.float32); m[10, 7] = 2.0; m[40, 30] = 2.0
    maps.append(m); names.append("exact_tie_first_wins")
    m = gauss(15.0, 20.0) + gauss(30.0, 45.0); m[45, 30] = m[20, 15]
    maps.append(m.astype(np.float32)); names.append("tie_two_blobs")
    # clamp plateau: the raw maximum sits in a region whose blurred value is negative while the blurred
    # maximum is positive -> every tap clamps to 1e-10, L is flat, det == 0, no refinement (pose_metrics.py:94)
    m = np.zeros((H, W), np.float32)
    m[24:37, 14:27] = -5.0
    m[30, 20] = 10.0
    m[45:60, 28:43] = 5.0
    maps.append(m); names.append("clamp_plateau_det0")
    maps.append(gauss(23.37, 31.81, 0.9) + 0.02); names.append("gauss_offset_bg")
    maps.append(gauss(24.5, 32.5, 3.0)); names.append("gauss_half_px")
    m = np.full((H, W), 0.5, np.float32)
    maps.append(m); names.append("constant_positive")
    return np.stack(maps), names


def gen_decode(ns, net_maps):
    pm = ns.pose_metrics
    gt = pm.GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)
    basic = pm.BasicKeyPointDecoder()
    enc = ns.transforms.RefineSimpleTransform.get_heat_map
    out = {}

    # (i) encoder-generated Gaussian maps + 1e-3 absolute noise, B=2
    joints = synth.joints_batch(2, 17, seed=11)
    joints[..., 0] = np.clip(joints[..., 0], 1.0, 46.5)
    joints[..., 1] = np.clip(joints[..., 1], 1.0, 62.5)
    joints[..., 2] = 1.0
    g = np.stack([enc(joints[b], 2.0, (48, 64))[0] for b in range(2)])
    g = g + synth.tensor_normal(12, "decode/noise", g.shape, std=1e-3)
    g = g.astype(np.float32)
    out["gauss/joints"] = joints
    sets = {"gauss": g, "net": net_maps.astype(np.float32)}
    # (ii) noise-like maps are regenerated from synth in the tests (bit-exact), only outputs stored
    sets["noise"] = synth.tensor_normal(13, "decode/noise_maps", (4, 17, 64, 48), std=1.0)
    # (iii) edge cases, padded to a multiple of 17 joints with a benign Gaussian
    em, names = edge_maps()
    pad = (-len(em)) % 17
    filler = em[names.index("gauss_half_px")]
    em = np.concatenate([em, np.repeat(filler[None], pad, 0)]).reshape(-1, 17, 64, 48)
    sets["edge"] = em
    out["edge/names"] = np.array(names)
    for tag, maps in sets.items():
        B = maps.shape[0]
        for tname, tinv in (("ident4", synth.trans_inv_batch(B)), ("rand", synth.trans_inv_batch(B, seed=21))):
            hm_t = torch.from_numpy(maps.copy())
            kps, mv = gt(hm_t, torch.from_numpy(tinv))
            assert torch.equal(hm_t, torch.from_numpy(maps)), "decoder mutated its input"
            out[f"{tag}/{tname}/gt_kps"] = kps.numpy()
            out[f"{tag}/{tname}/gt_max"] = mv.numpy()
            kps, mv = basic(torch.from_numpy(maps.copy()), torch.from_numpy(tinv))
            out[f"{tag}/{tname}/basic_kps"] = kps.numpy()
        co, mv = pm.BasicKeyPointDecoder.heat_map_to_axis(torch.from_numpy(maps.copy()))
        out[f"{tag}/axis"] = co.numpy()
        out[f"{tag}/axis_max"] = mv.numpy()
        if tag not in ("noise", "net"):
            out[f"{tag}/maps"] = maps
        # the reference's blurred+log map is not exposed; store its raw blur for the bit-exactness pin
        blur = torch.nn.functional.conv2d(torch.from_numpy(maps), gt.blur_weights, None, 1, 5, groups=17)
        out[f"{tag}/blur_max"] = blur.view(B, 17, -1).max(-1)[0].numpy()
    out["blur_weights"] = gt.blur_weights[0, 0].numpy()
    np.savez_compressed(os.path.join(GOLD, "g4_decode.npz"), **out)
    print("g4_decode.npz", {k: v.shape for k, v in out.items() if k.endswith("gt_kps")})


def gen_encode(ns):
    refine = ns.transforms.RefineSimpleTransform.get_heat_map
    basic = ns.transforms.BasicSimpleTransform.get_heat_map
    # heat-map-px joints: interior fractional, out of range, trunc-toward-zero band (-7,-6], vis=0
    special = np.array([
        [23.37, 31.81, 1], [0.0, 0.0, 1], [47.0, 63.0, 1], [-3.2, 10.5, 1], [50.9, 66.2, 1],
        [-6.5, 20.0, 1],    # ul=int(-12.5)=-12, br=int(-0.5+1)=0 -> inside
        [-7.5, 20.0, 1],    # br=int(-1.5+1)=int(-0.5)=0 -> inside only because int() truncates toward zero
        [-8.5, 20.0, 1],    # br=int(-2.5+1)=int(-1.5)=-1 -> outside, weight 0
        [-9.5, 20.0, 1],    # br=-2 -> outside
        [54.2, 20.0, 1],    # ul=int(48.2)=48 >= W -> outside
        [53.9, 20.0, 1],    # ul=47 -> inside
        [20.0, 70.3, 1], [20.0, 69.9, 1], [20.0, -9.01, 1],
        [12.25, 40.75, 0], [100.0, 100.0, 0], [30.5, 30.5, 0.5], [30.5, 30.5, 0.6],
    ], dtype=np.float32)
    rnd = synth.joints_batch(1, 16, seed=31)[0]
    joints = np.concatenate([special, rnd]).astype(np.float32)  # [34,3]
    t, w = refine(joints.copy(), 2.0, (48, 64))
    out = {"refine/joints": joints, "refine/targets": t, "refine/weights": w}
    jb = joints.copy()
    jb[:, :2] *= 4.0  # Basic variant takes INPUT px (stride 4)
    jb[5:9, 0] = np.array([-26.0, -30.1, -33.9, -38.0], np.float32)
    t, w = basic(jb.copy(), 2.0, (48, 64), 4)
    out.update({"basic/joints": jb, "basic/targets": t, "basic/weights": w})
    np.savez_compressed(os.path.join(GOLD, "g5_encode.npz"), **out)
    print("g5_encode.npz", t.shape, "refine weights", out["refine/weights"][:18])


def gen_nms(ns):
    """g8_nms.npz: SURVEY 8(f)4 - eval.py:153-197 (per-image rescoring + OKS-NMS through the reference's own function, run in
    a scratch directory with COCOeval switched off), direct oks_nms calls with in_vis_thresh / custom sigmas, and
    kps_to_dict_'s score rule (metrics/pose_metrics.py:172-179)."""
    import importlib, json, tempfile
    ev = importlib.import_module("eval")
    nd = importlib.import_module("datasets.naive_data")
    rng = np.random.default_rng(8)
    sizes, ids = [9, 1, 7, 6, 30], [139, 285, 632, 724, 785]
    kps_all, box, area, img = [], [], [], []
    for n, iid in zip(sizes, ids):
        base = rng.random((max(1, n // 3), 17, 3)) * np.array([400.0, 600.0, 1.0])
        pick = rng.integers(0, len(base), n)
        k = base[pick] + rng.normal(size=(n, 17, 3)) * np.array([rng.choice([0.5, 4.0]), rng.choice([0.5, 4.0]), 0.04])
        k[..., 2] = np.clip(k[..., 2], 0.0, 1.0)
        kps_all.append(k.astype(np.float32)); box.append(rng.random(n)); img += [iid] * n
        area.append((rng.random(n) * 30000 + 800).astype(np.float32))
    kps_all, box, area = np.concatenate(kps_all), np.concatenate(box), np.concatenate(area)
    kps_all[3, :, 2] = 0.1                                   # a person without a visible joint -> score 0
    kps_all[5, :, 2] = 0.15
    out = {"kps": kps_all, "box_score": box, "area": area, "img_id": np.array(img)}
    cwd = os.getcwd()
    ev.eval_kps = lambda *a, **k: None
    for tag, (vis, thr) in {"a": (0.2, 0.9), "b": (0.2, 0.5), "c": (0.5, 0.7)}.items():
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                items = [{"kps": kp.reshape(-1).tolist(), "area": float(a), "score": float(b), "img_id": int(i)}
                         for kp, a, b, i in zip(kps_all, area, box, img)]                    # as eval.py:139-147 builds them
                with open("predicts_kps_temp.json", "w") as wf:
                    json.dump(items, wf)
                ev.temp_read_in_and_filter(in_vis_thre=vis, oks_thre=thr)
                with open("filter_kps_predicts.json") as rf:
                    res = json.load(rf)
            finally:
                os.chdir(cwd)
        out[f"filter/{tag}/params"] = np.array([vis, thr])
        out[f"filter/{tag}/image_id"] = np.array([r["image_id"] for r in res])
        out[f"filter/{tag}/score"] = np.array([r["score"] for r in res], np.float64)
        out[f"filter/{tag}/keypoints"] = np.array([r["keypoints"] for r in res], np.float64)
    grp = slice(sum(sizes[:4]), sum(sizes))                 # the 30-person image, called directly
    sc = rng.random(30)
    sig = rng.random(17) * 0.1 + 0.02
    out["direct/scores"], out["direct/sigmas"] = sc, sig
    k64, a64 = kps_all[grp].astype(np.float64), area[grp].astype(np.float64)
    out["direct/keep_vis"] = np.array(nd.oks_nms(k64, sc, a64, 0.6, None, 0.3), np.int64)
    out["direct/keep_sig"] = np.array(nd.oks_nms(k64, sc, a64, 0.8, sig, None), np.int64)
    out["direct/oks_row"] = nd.oks_iou(k64[0], k64[1:], a64[0], a64[1:], None, None)
    out["direct/oks_row_vis"] = nd.oks_iou(k64[0], k64[1:], a64[0], a64[1:], None, 0.3)
    # kps_to_dict_
    pm = ns.pose_metrics
    pred = torch.from_numpy(kps_all[:8, :, :2].copy())
    mv = torch.from_numpy(kps_all[:8, :, 2:].copy())
    lst = []
    pm.kps_to_dict_(pred, mv, [int(i) for i in img[:8]], lst)
    out["dict/score"] = np.array([d["score"] for d in lst], np.float64)
    out["dict/keypoints"] = np.array([d["keypoints"] for d in lst], np.float64)
    out["dict/image_id"] = np.array([d["image_id"] for d in lst])
    np.savez_compressed(os.path.join(GOLD, "g8_nms.npz"), **out)
    print("g8_nms.npz kept", {t: len(out[f"filter/{t}/score"]) for t in "abc"}, "of", len(kps_all), "direct", len(out["direct/keep_vis"]),
          len(out["direct/keep_sig"]))


def gen_crop(ns):
    """g9_crop.npz: SURVEY 8(f)3 - the reference's BasicTransform.__call__ (datasets/naive_data.py:33-56: box -> centre/scale ->
    get_affine_transform -> cv.warpAffine) on a synthetic image.  cv2 is absent here: its two primitives are the restatements of
    oracle/pose_oracle.* plugged into the cv2 stub, so this fixture pins the reference's GLUE (conventions, float32 point
    construction, which matrix goes where), not OpenCV's arithmetic."""
    import importlib
    from scipy import ndimage
    nd = importlib.import_module("datasets.naive_data")
    rng = np.random.default_rng(9)
    img = ndimage.gaussian_filter(rng.random((240, 320, 3)) * 255, (2, 2, 0)).astype(np.uint8)
    boxes = np.array([[40.3, 30.7, 140.9, 200.2], [-20.0, 10.0, 90.0, 120.0], [200.0, 100.0, 330.0, 250.0], [100.0, 50.0, 110.0, 230.0],
                      [10.0, 10.0, 300.0, 60.0]], np.float32)
    tf = nd.BasicTransform()
    crops, tinv, centers, scales, areas = [], [], [], [], []
    for b in boxes:
        item = nd.KeyPointItem("x.jpg", b, 0.9)
        item.img = img
        item = tf(item)
        crops.append(item.img); tinv.append(item.trans_inv); centers.append(item.center); scales.append(item.scale); areas.append(item.area)
    np.savez_compressed(os.path.join(GOLD, "g9_crop.npz"), img=img, boxes=boxes, crops=np.stack(crops), trans_inv=np.stack(tinv),
                        centers=np.stack(centers), scales=np.stack(scales), areas=np.array(areas))
    print("g9_crop.npz", np.stack(crops).shape, "mean", np.stack(crops).mean())


WIDE_W_SEED, WIDE_X_SEED = 1, 7


def gen_forward_wide(ns):
    """g10_fwd_wide.npz (round 4: more DISTINCT reference inputs than G1-G3's one or two images, other weights too): eval forward of the
    real reference on 8 images (ResNet50-DConv, -DUC) / 4 images (HRNet-W32, DConv with SELayer), conditioned weights of seed 1, inputs of
    seed 7.  To stay small the heat maps are stored sub-sampled (every 4th row and column: [B,17,16,12] fp32) next to per-(image, joint)
    sum / L2 norm (fp64 of the fp32 maps), maximum and arg-max, plus the reference's own GaussTaylor key points on the full maps."""
    torch.set_num_threads(8)
    pm = ns.pose_metrics
    gt = pm.GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)
    out = {"w_seed": WIDE_W_SEED, "x_seed": WIDE_X_SEED}
    nets = (("dconv", lambda: ns.dconv.resnet50(pretrained=False, num_classes=17), 8),
            ("duc", lambda: ns.duc.resnet50(pretrained=False, num_classes=17), 8),
            ("hrnet_w32", lambda: ns.hrnet.get_pose_net(ns.hrnet_w32_yaml, pretrained=None, joint_num=17), 4),
            ("dconv_se", lambda: ns.dconv.resnet50(pretrained=False, num_classes=17, reduction=True), 4))
    for tag, make, B in nets:
        net = make()
        synth.load_conditioned(net, WIDE_W_SEED)
        net.eval()
        x = torch.from_numpy(synth.input_images(B, WIDE_X_SEED))
        with torch.no_grad():
            hm = net(x)
        kps, mv = gt(hm.clone(), torch.from_numpy(synth.trans_inv_batch(B)))
        h = hm.numpy()
        flat = h.reshape(B, 17, -1)
        out[f"{tag}/heat_sub"] = h[:, :, ::4, ::4].copy()
        out[f"{tag}/heat_sum"] = flat.astype(np.float64).sum(-1)
        out[f"{tag}/heat_l2"] = np.sqrt((flat.astype(np.float64) ** 2).sum(-1))
        out[f"{tag}/heat_max"] = flat.max(-1)
        out[f"{tag}/heat_argmax"] = flat.argmax(-1).astype(np.int64)
        out[f"{tag}/gt_kps"] = kps.numpy()
        out[f"{tag}/gt_max"] = mv.numpy()
        print("g10", tag, h.shape, "absmax", np.abs(h).max(), "std", h.std())
    np.savez_compressed(os.path.join(GOLD, "g10_fwd_wide.npz"), **out)


VARIANTS = (("resnet18", "dconv", False), ("resnet34", "duc", False), ("wide_resnet50_2", "dconv", False), ("resnet18", "dconv", True))


def gen_variants(ns, fname="g11_resnet_variants.npz"):
    """g11_resnet_variants.npz (round 4): the reference's other ResNet factories (nets/pose_resnet_dconv.py:282-403, pose_resnet_duc.py) - the
    BasicBlock nets and a wide Bottleneck net, one with SELayers - eval forward on 2 images (weight seed 1, input seed 7): state_dict key
    lists + shapes, sub-sampled heat maps, per-joint sums / norms / arg-max, the reference's GaussTaylor key points."""
    torch.set_num_threads(8)
    gt = ns.pose_metrics.GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)
    out = {"w_seed": WIDE_W_SEED, "x_seed": WIDE_X_SEED}
    x = torch.from_numpy(synth.input_images(2, WIDE_X_SEED))
    for arch, head, se in VARIANTS:
        tag = f"{arch}_{head}" + ("_se" if se else "")
        net = getattr(ns.dconv if head == "dconv" else ns.duc, arch)(pretrained=False, num_classes=17, reduction=se)
        synth.load_conditioned(net, WIDE_W_SEED)
        net.eval()
        with torch.no_grad():
            hm = net(x)
        kps, mv = gt(hm.clone(), torch.from_numpy(synth.trans_inv_batch(2)))
        h = hm.numpy()
        flat = h.reshape(2, 17, -1)
        sd = net.state_dict()
        out[f"{tag}/keys"] = np.array(list(sd.keys()))
        out[f"{tag}/shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        out[f"{tag}/heat_sub"] = h[:, :, ::4, ::4].copy()
        out[f"{tag}/heat_sum"] = flat.astype(np.float64).sum(-1)
        out[f"{tag}/heat_l2"] = np.sqrt((flat.astype(np.float64) ** 2).sum(-1))
        out[f"{tag}/heat_max"] = flat.max(-1)
        out[f"{tag}/heat_argmax"] = flat.argmax(-1).astype(np.int64)
        out[f"{tag}/gt_kps"] = kps.numpy()
        print(fname[:3], tag, h.shape, "absmax", np.abs(h).max(), "std", h.std(), "keys", len(sd))
    np.savez_compressed(os.path.join(GOLD, fname), **out)


RESNEXT = (("resnext50_32x4d", "dconv", False), ("resnext101_32x8d", "duc", False))


def gen_resnext(ns):
    """g12_resnext.npz (round 5): the reference's grouped factories (nets/pose_resnet_dconv.py:342-368, `groups = 32` at :101), same content as g11."""
    global VARIANTS
    keep, VARIANTS = VARIANTS, RESNEXT
    try:
        gen_variants(ns, fname="g12_resnext.npz")
    finally:
        VARIANTS = keep


def main():
    assert ref_import.available(), "needs /root/reference (build container only)"
    os.makedirs(GOLD, exist_ok=True)
    ns = ref_import.load()
    if "--only-wide" in sys.argv:          # (round 4 additions; the other files regenerate bit for bit and were left as committed)
        gen_forward_wide(ns)
        return
    if "--only-variants" in sys.argv:
        gen_variants(ns)
        return
    if "--only-resnext" in sys.argv:
        gen_resnext(ns)
        return
    if "--only-train-b8" in sys.argv:      # (round 6 addition)
        gen_train_b8(ns)
        return
    hm = gen_forward(ns)  # returns the DUC maps last; reload dconv maps for the decoder set
    net_maps = np.load(os.path.join(GOLD, "g1_dconv_fwd.npz"))["heat_maps"]
    gen_decode(ns, net_maps)
    gen_encode(ns)
    gen_hrnet(ns)
    gen_se(ns)
    gen_next(ns)
    gen_train(ns)
    gen_train_b8(ns)
    gen_nms(ns)
    gen_crop(ns)
    gen_forward_wide(ns)
    gen_variants(ns)
    gen_resnext(ns)
    del hm


if __name__ == "__main__":
    main()
