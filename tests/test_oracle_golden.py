"""Pin the oracle (oracle/*) against the golden vectors frozen from the real reference
(tests/golden/*.npz, made by oracle/gen_golden.py).  CPU only; no reference needed."""
import numpy as np
import pytest
import torch

from oracle import nets_oracle, pose_oracle
from simple_pose_amd import synth


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2**31) - a, a)
    b = np.where(b < 0, np.int64(-2**31) - b, b)
    return np.abs(a - b)


def _decode_inputs(g4, tag):
    if tag == "noise":
        return synth.tensor_normal(13, "decode/noise_maps", (4, 17, 64, 48), std=1.0)
    if tag == "net":
        return None
    return g4[f"{tag}/maps"]


@pytest.mark.parametrize("tag", ["gauss", "noise", "edge", "net"])
def test_decoders_match_reference(golden, tag):
    g4 = golden("g4_decode.npz")
    maps = _decode_inputs(g4, tag)
    if maps is None:
        maps = golden("g1_dconv_fwd.npz")["heat_maps"]
    B = maps.shape[0]
    # heat_map_to_axis: bit exact (integer work)
    co, mv = pose_oracle.heat_map_to_axis(maps)
    assert np.array_equal(co, g4[f"{tag}/axis"])
    assert np.array_equal(mv, g4[f"{tag}/axis_max"])
    for tname, tinv in (("ident4", synth.trans_inv_batch(B)), ("rand", synth.trans_inv_batch(B, seed=21))):
        kps, mv = pose_oracle.decode_gauss_taylor(maps, tinv)
        ref = g4[f"{tag}/{tname}/gt_kps"]
        assert np.array_equal(mv, g4[f"{tag}/{tname}/gt_max"])
        err = np.abs(kps - ref)
        scale = np.abs(tinv[:, :, :2]).sum(-1).max()  # px in image space per heat-map px
        # BASELINE.json bar: decoded keypoints within 1e-3 px (heat-map px; image px scale with trans_inv).
        # Measured here: identical bits with the x4 trans_inv on Gaussian/edge maps, <= 1.7e-4 px on the
        # noise-like network maps (1-ulp torch.log / MKL inverse differences).
        assert err.max() <= 1e-3 * max(scale, 1.0), (tag, tname, err.max())
        if tag in ("gauss", "edge") and tname == "ident4":
            assert np.array_equal(kps, ref)
        bk, _ = pose_oracle.decode_basic(maps, tinv)
        berr = np.abs(bk - g4[f"{tag}/{tname}/basic_kps"])
        assert berr.max() <= 1e-4 * max(scale, 1.0) * 64, (tag, tname, berr.max())


def test_blur_kernel_and_blur_bit_exact(golden):
    g4 = golden("g4_decode.npz")
    assert np.array_equal(pose_oracle.blur_kernel(11), g4["blur_weights"])


def test_encoders_match_reference(golden):
    g5 = golden("g5_encode.npz")
    t, w = pose_oracle.encode_refine(g5["refine/joints"], 2.0, (48, 64))
    assert np.array_equal(w, g5["refine/weights"])
    assert _ulp_diff(t, g5["refine/targets"]).max() <= 1
    assert (t == g5["refine/targets"]).mean() > 0.999
    t, w = pose_oracle.encode_basic(g5["basic/joints"], 2.0, (48, 64), 4)
    assert np.array_equal(w, g5["basic/weights"])
    assert _ulp_diff(t, g5["basic/targets"]).max() <= 2
    assert np.array_equal(t != 0, g5["basic/targets"] != 0)


@pytest.mark.parametrize("arch,fname,head", [("resnet50_dconv", "g1_dconv_fwd.npz", "dconv"),
                                               ("resnet50_duc", "g2_duc_fwd.npz", "duc")])
def test_forward_oracle_matches_reference(golden, arch, fname, head):
    g = golden(fname)
    shapes = nets_oracle.state_dict_shapes_resnet50(head)
    assert len(shapes) == int(g["n_state_keys"])
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, int(g["seed"])).items()}
    x = torch.from_numpy(synth.input_images(int(g["batch"]), int(g["seed"])))
    taps = {}
    with torch.no_grad():
        hm = nets_oracle.FORWARDS[arch](sd, x, tap=lambda n, t: taps.__setitem__(n, t)).numpy()
    ref = g["heat_maps"]
    rel = np.abs(hm - ref).max() / np.abs(ref).max()
    assert rel <= 1e-5, rel  # same ATen/oneDNN arithmetic; only thread-blocking differences
    for name, t in taps.items():
        ref_slice = g[name + "/slice"]
        assert np.abs(t[0, :8, :4, :4].numpy() - ref_slice).max() <= 1e-4 * max(1.0, float(g[name + "/absmax"]))
        assert abs(t.double().mean().item() - float(g[name + "/mean"])) < 1e-5


def test_hrnet_forward_oracle_and_key_layout_match_reference(golden):
    import yaml
    from simple_pose_amd.nets.pose_hrnet import hrnet_state_dict_shapes, load_cfg
    import os
    g = golden("g3_hrnet_w32_fwd.npz")
    cfg = load_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple_pose_amd", "nets", "hrnet_w32.yaml"))
    shapes = hrnet_state_dict_shapes(cfg, 17)
    assert [k for k, _, _ in shapes] == list(g["keys"])                       # names AND order of the reference
    assert [",".join(str(d) for d in s) for _, s, _ in shapes] == list(g["shapes"])
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, int(g["seed"])).items()}
    x = torch.from_numpy(synth.input_images(1, int(g["seed"])))
    with torch.no_grad():
        hm = nets_oracle.hrnet_forward(sd, x, cfg).numpy()
    ref = g["heat_maps"]
    assert np.abs(hm - ref).max() / np.abs(ref).max() <= 1e-5


def test_train_step_oracle_matches_reference(golden):
    """oracle/train_oracle.py (functional forward + autograd + restated Adam) against one step of the real reference."""
    from oracle import train_oracle
    g = golden("g6_train_step.npz")
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv")
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, 0).items()}
    x = torch.from_numpy(synth.input_images(2, 0))
    t, w = pose_oracle.encode_refine(g["joints"], 2.0, (48, 64))
    loss, grads, heat = train_oracle.forward_backward(sd, x, torch.from_numpy(t), torch.from_numpy(w))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert np.abs(heat.numpy() - g["heat_train"]).max() <= 1e-4 * np.abs(g["heat_train"]).max()
    picks = [k[5:] for k in g.files if k.startswith("grad/")]
    for k in picks:
        ref = g["grad/" + k]
        got = grads[k].numpy()[tuple(slice(0, s) for s in ref.shape)]
        scale = float(g["gradnorm/" + k]) / np.sqrt(grads[k].numel())
        assert np.abs(got - ref).max() <= 2e-3 * scale + 1e-12, k
    params = {k: sd[k] for k in grads}
    train_oracle.adam_step(params, grads, {}, lr=1e-3)
    for k in picks:
        ref = g["param/" + k]
        got = params[k].numpy()[tuple(slice(0, s) for s in ref.shape)]
        assert np.abs(got - ref).max() <= 2e-5, k          # |update| = lr = 1e-3 on the first Adam step
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_var", "deconv_layers.7.running_mean"):
        assert np.abs(sd[k].numpy() - g["buf/" + k]).max() <= 1e-5 * max(1.0, np.abs(g["buf/" + k]).max()), k
    assert int(sd["bn1.num_batches_tracked"]) == int(g["buf/bn1.num_batches_tracked"]) == 1


def test_train_step_oracle_matches_reference_at_batch_8(golden):
    """G6b (round 6): one step of the real reference at B = 8, 256x192, pinned by quantities its own fp32 arithmetic holds to 1e-3 - the fp64
    norm and two sketches <grad, r> of EVERY parameter's gradient (oracle/gen_golden.py::gen_train_b8) - next to G6's slices.  The oracle
    (same torch ops, same thread count) reproduces them to rounding."""
    from oracle import train_oracle
    from oracle.train_oracle import gradient_sketch_vector
    g = golden("g6b_train_step_b8.npz")
    B = int(g["batch"])
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv")
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, int(g["seed"])).items()}
    x = torch.from_numpy(synth.input_images(B, int(g["seed"])))
    t, w = pose_oracle.encode_refine(g["joints"], 2.0, (48, 64))
    torch.set_num_threads(8)
    loss, grads, heat = train_oracle.forward_backward(sd, x, torch.from_numpy(t), torch.from_numpy(w))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert np.abs(heat.numpy()[:, :, ::4, ::4] - g["heat_train_sub"]).max() <= 1e-4 * np.abs(g["heat_train_sub"]).max()
    keys = [str(k) for k in g["keys"]]
    assert keys == list(grads) and len(keys) == 170
    worst_n = worst_s = 0.0
    for i, k in enumerate(keys):
        gd = grads[k].double()
        n = float(g["gradnorm"][i])
        worst_n = max(worst_n, abs(float(gd.norm()) - n) / n)
        for j in range(2):
            sk = float((gd.numpy() * gradient_sketch_vector(k, j, gd.shape)).sum())
            worst_s = max(worst_s, abs(sk - float(g["sketch"][i, j])) / n)
    # measured here: 0 (8 threads, the generator's count) ... 1.1e-3 / 8.4e-3 (1 thread): the reference's own summation-order spread
    assert worst_n <= 2e-3 and worst_s <= 1.5e-2, (worst_n, worst_s)
    for k in [k[5:] for k in g.files if k.startswith("grad/")]:
        ref = g["grad/" + k]
        got = grads[k].numpy()[tuple(slice(0, s) for s in ref.shape)]
        scale = float(g["gradnorm"][keys.index(k)]) / np.sqrt(grads[k].numel())
        assert np.abs(got - ref).max() <= 6e-2 * scale + 1e-12, k      # (a slice's max moves by 2-5e-2 with the thread count: tests/measure_reference_spread.py)


def test_se_variant_oracle_and_key_layout_match_reference(golden):
    g = golden("g1s_dconv_se_fwd.npz")
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv", se=True)
    assert [k for k, _, _ in shapes] == list(g["keys"])
    assert [",".join(str(d) for d in s) for _, s, _ in shapes] == list(g["shapes"])
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, int(g["seed"])).items()}
    with torch.no_grad():
        hm = nets_oracle.resnet_dconv_forward(sd, torch.from_numpy(synth.input_images(1, int(g["seed"])))).numpy()
    assert np.abs(hm - g["heat_maps"]).max() / np.abs(g["heat_maps"]).max() <= 1e-5


def test_next_rows_oracle_matches_reference(golden):
    """HeatMapAcc and the collate_fn normalisation."""
    g = golden("g7_next.npz")
    for tag in ("a", "b"):
        tgt, _ = pose_oracle.encode_refine(g[f"acc/{tag}/joints"], 2.0, (48, 64))
        assert abs(float(pose_oracle.heat_map_acc(g[f"acc/{tag}/pred"], tgt)) - float(g[f"acc/{tag}/value"])) < 1e-6
    assert np.array_equal(pose_oracle.normalize_crops(g["collate/img_u8"]), g["collate/input"])


def _filter_like_eval(g, vis, thr, rescore, nms):
    """eval.py:153-197 with the given rescoring / NMS functions; returns (image_id, score, keypoints) lists."""
    kps, box, area, img = g["kps"].astype(np.float64), g["box_score"], g["area"].astype(np.float64), g["img_id"]
    ids = list(dict.fromkeys(img.tolist()))
    out_id, out_sc, out_kp = [], [], []
    for iid in ids:
        rows = np.nonzero(img == iid)[0]
        sc = rescore(kps[rows], box[rows], vis)
        for r in nms(kps[rows], sc, area[rows], thr):
            out_id.append(iid); out_sc.append(sc[r]); out_kp.append(kps[rows][r].reshape(-1))
    return np.array(out_id), np.array(out_sc), np.array(out_kp)


def test_oks_nms_oracle_matches_reference(golden):
    g = golden("g8_nms.npz")
    for tag in "abc":
        vis, thr = g[f"filter/{tag}/params"]
        ids, sc, kp = _filter_like_eval(g, vis, thr, pose_oracle.pose_rescore, pose_oracle.oks_nms)
        np.testing.assert_array_equal(ids, g[f"filter/{tag}/image_id"])
        np.testing.assert_array_equal(sc, g[f"filter/{tag}/score"])              # float64, bit for bit
        np.testing.assert_array_equal(kp, g[f"filter/{tag}/keypoints"])
    k64 = g["kps"][23:].astype(np.float64)
    a64 = g["area"][23:].astype(np.float64)
    assert pose_oracle.oks_nms(k64, g["direct/scores"], a64, 0.6, None, 0.3) == g["direct/keep_vis"].tolist()
    assert pose_oracle.oks_nms(k64, g["direct/scores"], a64, 0.8, g["direct/sigmas"], None) == g["direct/keep_sig"].tolist()
    # equal scores: deterministic "higher index first" (the reference's order for ties is whatever numpy's argsort leaves)
    same = np.zeros(5)
    far = np.arange(5)[:, None, None] * 1000.0 + np.zeros((5, 17, 3))
    assert pose_oracle.oks_nms(far, same, np.full(5, 1e3), 0.9) == [4, 3, 2, 1, 0]
    np.testing.assert_allclose(pose_oracle.pose_score(g["kps"][:8, :, 2]), g["dict/score"], rtol=2e-7)


def test_crop_geometry_and_warp_restatement(golden):
    """SURVEY 8(f)3.  (1) the host-side geometry mirror reproduces the matrices the reference's BasicTransform produced; (2) the
    restated cv.warpAffine reproduces the crops frozen through the reference's glue; (3) absent cv2, the restatement itself is only
    sanity-bounded: within one grey level of exact (float64) bilinear interpolation on a smooth image ("parity unpinned")."""
    from scipy import ndimage
    from simple_pose_amd.commons.joint_utils import box_to_center_scale, get_affine_transform
    g = golden("g9_crop.npz")
    img = g["img"]
    for i, (x1, y1, x2, y2) in enumerate(g["boxes"]):
        c, s = box_to_center_scale(x1, y1, x2 - x1, y2 - y1, 192 / 256)
        np.testing.assert_array_equal(c, g["centers"][i]); np.testing.assert_array_equal(s, g["scales"][i])
        m, _ = get_affine_transform(c, s, 0, (192, 256))
        _, tinv = get_affine_transform(c, s, 0, (48, 64))
        np.testing.assert_array_equal(tinv, g["trans_inv"][i])
        crop = pose_oracle.warp_affine_u8c3(img, m, (192, 256))
        np.testing.assert_array_equal(crop, g["crops"][i])
        mi = np.linalg.inv(np.vstack([m, [0, 0, 1]]))[:2]
        ys, xs = np.mgrid[0:256, 0:192]
        sx, sy = mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2], mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]
        ref = np.stack([ndimage.map_coordinates(img[..., k].astype(float), [sy, sx], order=1, mode="constant", cval=0) for k in range(3)], -1)
        inside = (sx > 0) & (sx < img.shape[1] - 1) & (sy > 0) & (sy < img.shape[0] - 1)
        assert np.abs(crop.astype(float) - ref)[inside].max() <= 1.0
        assert (crop[(sx < -1) | (sy < -1) | (sx > img.shape[1]) | (sy > img.shape[0])] == 0).all()      # BORDER_CONSTANT 0


WIDE_NETS = [("dconv", 8), ("duc", 8), ("hrnet_w32", 4), ("dconv_se", 4)]


def _wide_state_dict(tag, seed):
    """(state_dict of torch tensors, forward callable) of one g10 net - shapes from the build's own tables."""
    import os
    if tag == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import hrnet_state_dict_shapes, load_cfg
        cfg = load_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple_pose_amd", "nets", "hrnet_w32.yaml"))
        sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(hrnet_state_dict_shapes(cfg, 17), seed).items()}
        return sd, lambda sd_, x: nets_oracle.hrnet_forward(sd_, x, cfg)
    head = "duc" if tag == "duc" else "dconv"
    shapes = nets_oracle.state_dict_shapes_resnet50(head, se=True) if tag == "dconv_se" else nets_oracle.state_dict_shapes_resnet50(head)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed).items()}
    return sd, nets_oracle.FORWARDS["resnet50_" + head]


@pytest.mark.parametrize("tag,B", WIDE_NETS, ids=[t for t, _ in WIDE_NETS])
def test_forward_oracle_matches_reference_on_the_wide_set(golden, tag, B):
    """g10_fwd_wide.npz (round 4): 8 / 8 / 4 / 4 DISTINCT images through the real reference with a second set of weights - the forward
    oracle reproduces the sub-sampled maps, the per-joint sums / norms / maxima and arg-max cells, and the C decoder the key points."""
    g = golden("g10_fwd_wide.npz")
    sd, fwd = _wide_state_dict(tag, int(g["w_seed"]))
    x = torch.from_numpy(synth.input_images(B, int(g["x_seed"])))
    with torch.no_grad():
        hm = fwd(sd, x).numpy()
    ref = g[f"{tag}/heat_sub"]
    scale = np.abs(g[f"{tag}/heat_max"]).max()
    assert np.abs(hm[:, :, ::4, ::4] - ref).max() / scale <= 1e-5
    flat = hm.reshape(B, 17, -1)
    assert np.abs(flat.astype(np.float64).sum(-1) - g[f"{tag}/heat_sum"]).max() <= 1e-5 * scale * flat.shape[-1] ** 0.5
    assert np.abs(np.sqrt((flat.astype(np.float64) ** 2).sum(-1)) - g[f"{tag}/heat_l2"]).max() <= 1e-5 * g[f"{tag}/heat_l2"].max()
    assert np.abs(flat.max(-1) - g[f"{tag}/heat_max"]).max() <= 1e-5 * scale
    assert (flat.argmax(-1) == g[f"{tag}/heat_argmax"]).mean() >= 0.99
    kps, mv = pose_oracle.decode_gauss_taylor(hm, synth.trans_inv_batch(B))
    err = np.abs(kps - g[f"{tag}/gt_kps"]).max(-1) / 4.0
    assert (err <= 1e-3).mean() >= 0.9, (err <= 1e-3).mean()        # noise-like maps: the tail is the ill-conditioned -H^-1 g


VARIANTS = [("resnet18", "dconv", False), ("resnet34", "duc", False), ("wide_resnet50_2", "dconv", False), ("resnet18", "dconv", True),
            ("resnext50_32x4d", "dconv", False), ("resnext101_32x8d", "duc", False)]      # (round 5: grouped factories, g12_resnext.npz)


@pytest.mark.parametrize("arch,head,se", VARIANTS, ids=[f"{a}_{h}" + ("_se" if s else "") for a, h, s in VARIANTS])
def test_forward_oracle_matches_reference_on_the_resnet_variants(golden, arch, head, se):
    """g11_resnet_variants.npz: the BasicBlock nets, a wide Bottleneck net and an SELayer variant through the real reference - the forward oracle
    (driven by the state_dict alone) reproduces the sub-sampled maps, norms and arg-max cells."""
    from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc
    g = golden("g12_resnext.npz" if arch.startswith("resnext") else "g11_resnet_variants.npz")
    tag = f"{arch}_{head}" + ("_se" if se else "")
    m = getattr(pose_resnet_dconv if head == "dconv" else pose_resnet_duc, arch)(pretrained=False, num_classes=17, reduction=se)
    layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(layout, int(g["w_seed"])).items()}
    x = torch.from_numpy(synth.input_images(2, int(g["x_seed"])))
    with torch.no_grad():
        hm = nets_oracle.FORWARDS["resnet50_" + head](sd, x).numpy()
    scale = np.abs(g[f"{tag}/heat_max"]).max()
    assert np.abs(hm[:, :, ::4, ::4] - g[f"{tag}/heat_sub"]).max() / scale <= 1e-5
    flat = hm.reshape(2, 17, -1)
    assert np.abs(np.sqrt((flat.astype(np.float64) ** 2).sum(-1)) - g[f"{tag}/heat_l2"]).max() <= 1e-5 * g[f"{tag}/heat_l2"].max()
    assert (flat.argmax(-1) == g[f"{tag}/heat_argmax"]).mean() >= 0.97
