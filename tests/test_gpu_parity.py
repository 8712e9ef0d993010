"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the C ABI, against
(a) the golden vectors frozen from the real reference and (b) the oracle on seeded inputs.  Tolerances are the
ones BASELINE.json states: heat maps within 1e-4 relative (fp32), decoded key points within 1e-3 px."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_oracle, pose_oracle  # noqa: E402  (the checker)
from simple_pose_amd import _lib, engine, synth  # noqa: E402
from simple_pose_amd.commons.transforms import BasicSimpleTransform, RefineSimpleTransform  # noqa: E402
from simple_pose_amd.metrics import BasicKeyPointDecoder, GaussTaylorKeyPointDecoder  # noqa: E402
from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc  # noqa: E402
from tests.desc_interp import TorchPacker  # noqa: E402  (the packed layouts restated in torch: the checker of the pack kernels)

DEV = "cuda:0"


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2**31) - a, a)
    b = np.where(b < 0, np.int64(-2**31) - b, b)
    return np.abs(a - b)


# ---------------------------------------------------------------------------------------------- decoders
# px per unit of |trans_inv|, pinned from gpurun_out/measured_parity.json (tests/conftest.py `measured`), round 2: Basic decoder vs
# the reference 6.9e-6 (fp32 ulps of a 64-px coordinate), GaussTaylor vs the oracle on identical maps 0, vs the reference 3.8e-5
BASIC_BAR = 2e-5
GT_ORACLE_BAR = 2e-5
GT_REFERENCE_BAR = 1.2e-4  # (BASELINE contract: 1e-3 px)


def _decode_inputs(golden, tag):
    g4 = golden("g4_decode.npz")
    if tag == "noise":
        return synth.tensor_normal(13, "decode/noise_maps", (4, 17, 64, 48), std=1.0)
    if tag == "net":
        return golden("g1_dconv_fwd.npz")["heat_maps"]
    return g4[f"{tag}/maps"]


@pytest.mark.parametrize("tag", ["gauss", "noise", "edge", "net"])
def test_decoders_vs_reference_golden_and_oracle(golden, measured, tag):
    g4 = golden("g4_decode.npz")
    maps = _decode_inputs(golden, tag)
    B = maps.shape[0]
    hm = _cuda(maps)
    hm_before = hm.clone()
    co, mv = BasicKeyPointDecoder.heat_map_to_axis(hm)
    assert np.array_equal(co.cpu().numpy(), g4[f"{tag}/axis"])          # index work: bit exact
    assert np.array_equal(mv.cpu().numpy(), g4[f"{tag}/axis_max"])
    gt = GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)
    basic = BasicKeyPointDecoder()
    for tname, tinv in (("ident4", synth.trans_inv_batch(B)), ("rand", synth.trans_inv_batch(B, seed=21))):
        kps, mv = gt(hm, _cuda(tinv))
        kps, mv = kps.cpu().numpy(), mv.cpu().numpy()
        scale = max(np.abs(tinv[:, :, :2]).sum(-1).max(), 1.0)
        ref = g4[f"{tag}/{tname}/gt_kps"]
        assert kps.shape == ref.shape and mv.shape == (B, 17, 1)
        assert np.array_equal(mv, g4[f"{tag}/{tname}/gt_max"])
        measured(f"{tname}/gauss_taylor_vs_reference_px_over_scale", np.abs(kps - ref).max() / scale, GT_REFERENCE_BAR)
        assert np.abs(kps - ref).max() <= GT_REFERENCE_BAR * scale, (tag, tname, np.abs(kps - ref).max())
        okps, omv = pose_oracle.decode_gauss_taylor(maps, tinv)
        assert np.array_equal(mv, omv)
        measured(f"{tname}/gauss_taylor_vs_oracle_px_over_scale", np.abs(kps - okps).max() / scale, GT_ORACLE_BAR)
        assert np.abs(kps - okps).max() <= GT_ORACLE_BAR * scale, (tag, tname, np.abs(kps - okps).max())
        bk, _ = basic(hm, _cuda(tinv))
        # coordinates reach |trans_inv| * 64 px: one fp32 ulp there is 64 * 2^-24 * scale = 4e-6 * scale; the kernel's fp64 dot vs
        # the reference's fp32 einsum differ by a few such ulps
        berr = np.abs(bk.cpu().numpy() - g4[f"{tag}/{tname}/basic_kps"]).max()
        measured(f"{tname}/basic_vs_reference_px_over_scale", berr / scale, BASIC_BAR)
        assert berr <= BASIC_BAR * scale, (tag, tname, berr)
    assert torch.equal(hm, hm_before), "decoder must not modify its input"


def test_decoder_other_shapes_and_kernel_sizes():
    """Generic (non 64x48 / non ks=11) path against the oracle."""
    for (B, J, H, W, ks) in [(2, 5, 32, 24, 7), (1, 3, 40, 52, 11), (3, 17, 64, 48, 5), (2, 4, 17, 13, 3)]:
        maps = synth.tensor_normal(5, f"dec/{H}x{W}", (B, J, H, W), std=1.0)
        # add a Gaussian bump so that refinement is well conditioned for half of the joints
        yy, xx = np.mgrid[0:H, 0:W]
        for b in range(B):
            for j in range(0, J, 2):
                cx, cy = 3 + (7 * j + 3 * b) % (W - 6), 3 + (5 * j + b) % (H - 6)
                maps[b, j] = 8 * np.exp(-((xx - cx - 0.3) ** 2 + (yy - cy + 0.2) ** 2) / 8.0) + 0.01 * maps[b, j]
        tinv = synth.trans_inv_batch(B, seed=4)
        kps, mv = GaussTaylorKeyPointDecoder(ks, J)(_cuda(maps), _cuda(tinv))
        okps, omv = pose_oracle.decode_gauss_taylor(maps, tinv, ks)
        scale = max(np.abs(tinv[:, :, :2]).sum(-1).max(), 1.0)
        assert np.array_equal(mv.cpu().numpy(), omv)
        err = np.abs(kps.cpu().numpy() - okps).max(-1)
        assert (err <= 1e-3 * scale).mean() >= 0.9 and np.median(err) <= 1e-4 * scale, (H, W, ks, err.max())


# ---------------------------------------------------------------------------------------------- encoders
def test_encoders_vs_reference_golden(golden):
    g5 = golden("g5_encode.npz")
    t, w = RefineSimpleTransform.get_heat_map(g5["refine/joints"], 2.0, (48, 64))      # numpy in -> numpy out
    assert isinstance(t, np.ndarray) and t.shape == (34, 64, 48) and t.dtype == np.float32
    assert np.array_equal(w, g5["refine/weights"])
    assert _ulp_diff(t, g5["refine/targets"]).max() <= 1
    assert (t == g5["refine/targets"]).mean() > 0.999
    t, w = BasicSimpleTransform.get_heat_map(g5["basic/joints"], 2.0, (48, 64), 4)
    assert np.array_equal(w, g5["basic/weights"])
    assert _ulp_diff(t, g5["basic/targets"]).max() <= 2
    assert np.array_equal(t != 0, g5["basic/targets"] != 0)


def test_encoder_batched_vs_oracle_and_ragged_shapes():
    for (B, J, H, W) in [(8, 17, 64, 48), (3, 5, 30, 22), (1, 1, 7, 5)]:
        j = synth.joints_batch(B, J, seed=9, w=W, h=H)
        t, w = RefineSimpleTransform.get_heat_map(_cuda(j), 2.0, (W, H))
        ot, ow = pose_oracle.encode_refine(j, 2.0, (W, H))
        assert t.shape == (B, J, H, W) and t.is_cuda
        assert np.array_equal(w.cpu().numpy(), ow)
        assert _ulp_diff(t.cpu().numpy(), ot).max() <= 1
        jb = j.copy(); jb[..., :2] *= 4
        t, w = BasicSimpleTransform.get_heat_map(_cuda(jb), 2.0, (W, H), 4)
        ot, ow = pose_oracle.encode_basic(jb, 2.0, (W, H), 4)
        assert np.array_equal(w.cpu().numpy(), ow)
        assert _ulp_diff(t.cpu().numpy(), ot).max() <= 1


def test_encode_decode_round_trip_full_batch():
    """Size-independent property at BASELINE batch (128): decode(encode(j)) == j.  Joints >= 8 px from the border
    (the 11x11 blur's zero padding biases the Taylor step closer in): < 5e-3 px; and the HIP round trip equals the
    oracle's round trip."""
    B = 128
    j = synth.joints_batch(B, 17, seed=77)
    j[..., 0] = np.clip(j[..., 0], 8.0, 39.0)
    j[..., 1] = np.clip(j[..., 1], 8.0, 55.0)
    j[..., 2] = 1.0
    t, w = RefineSimpleTransform.get_heat_map(_cuda(j), 2.0, (48, 64))
    ident = np.zeros((B, 2, 3), np.float32); ident[:, 0, 0] = 1; ident[:, 1, 1] = 1
    kps, mv = GaussTaylorKeyPointDecoder()(t, _cuda(ident))
    err = np.abs(kps.cpu().numpy() - j[..., :2])
    assert err.max() < 5e-3 and np.median(err) < 1e-3, (err.max(), np.median(err))
    assert torch.all(w == 1)
    ot, _ = pose_oracle.encode_refine(j[:8], 2.0, (48, 64))
    okps, _ = pose_oracle.decode_gauss_taylor(ot, ident[:8])
    assert np.abs(kps[:8].cpu().numpy() - okps).max() <= 1e-4


# what bf16 heat maps (BASELINE configs 3 and 5) do to decoded key points, measured with the oracle on this very input (round 5): the bf16
# rounding of a sigma = 2 Gaussian target moves the GaussTaylor result by max 2.27e-3 / p99 1.70e-3 / median 1.3e-4 heat-map px (90.8 % of the
# coordinates stay within 1e-3 px); the bars are 3x the measurement.  (On the network's own noise-like maps the 1e-3 px contract is not
# meaningful even in fp32 - SURVEY App. E - so Gaussian-like maps are where this is stated.)
BF16_KP_MAX_BAR = 7e-3
BF16_KP_P99_BAR = 5.2e-3


def test_bf16_rounded_heat_maps_keep_key_points(measured):
    """encoder targets (sigma = 2, 128 x 17 joints >= 8 px from the border) -> cast to bf16 and back (what a bf16 network hands the decoder:
    it decodes the fp32 cast of bf16 maps) -> HIP GaussTaylor decode, against the ORACLE's decode of the fp32 maps
    (metrics/pose_metrics.py:62-107).  Also: on the identical rounded maps HIP and oracle agree as on any other map."""
    B = 128
    j = synth.joints_batch(B, 17, seed=77)
    j[..., 0] = np.clip(j[..., 0], 8.0, 39.0)
    j[..., 1] = np.clip(j[..., 1], 8.0, 55.0)
    j[..., 2] = 1.0
    t, _ = RefineSimpleTransform.get_heat_map(_cuda(j), 2.0, (48, 64))
    ident = np.zeros((B, 2, 3), np.float32); ident[:, 0, 0] = 1; ident[:, 1, 1] = 1
    t16 = t.to(torch.bfloat16).to(torch.float32)
    kps16, _ = GaussTaylorKeyPointDecoder()(t16, _cuda(ident))
    ot, _ = pose_oracle.encode_refine(j, 2.0, (48, 64))
    ok32, _ = pose_oracle.decode_gauss_taylor(ot, ident)
    d = np.abs(kps16.cpu().numpy() - ok32)
    measured("bf16_maps_kp_shift_max_px", d.max(), BF16_KP_MAX_BAR)
    measured("bf16_maps_kp_shift_p99_px", np.percentile(d, 99), BF16_KP_P99_BAR)
    measured("bf16_maps_kp_fraction_within_1e-3_px", (d < 1e-3).mean())
    assert d.max() <= BF16_KP_MAX_BAR and np.percentile(d, 99) <= BF16_KP_P99_BAR, (d.max(), np.percentile(d, 99))
    ok16, _ = pose_oracle.decode_gauss_taylor(t16.cpu().numpy(), ident)
    assert np.abs(kps16.cpu().numpy() - ok16).max() <= GT_ORACLE_BAR
    # the true joints: the fp32 round trip sits at 3.4e-4 px, the bf16 one inside the same 7e-3
    assert np.abs(kps16.cpu().numpy() - j[..., :2]).max() <= BF16_KP_MAX_BAR


# ---------------------------------------------------------------------------------------------- conv family
def _run_conv(x_nchw, builder_fn):
    """Build a one-layer program around an NHWC activation and run it on the GPU."""
    B, C, H, W = x_nchw.shape
    b = engine.ProgramBuilder(H, W)
    b.p.shapes["input"] = (H, W, C)
    out = builder_fn(b, "input")
    prog = b.p
    prog.out_name = out
    h, w, c = prog.shapes[out]
    prog.out_shape = (h, w, c)          # NHWC result for unit tests
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = prog.run(x)
    torch.cuda.synchronize()
    return y.cpu()


CONV_CASES = [
    # name, B, Cin, H, W, Cout, k, stride, pad
    ("1x1_64_256", 2, 64, 16, 12, 256, 1, 1, 0),
    ("1x1_256_64", 3, 256, 16, 12, 64, 1, 1, 0),
    ("3x3_s1_64", 2, 64, 16, 12, 64, 3, 1, 1),
    ("3x3_s2_128", 2, 128, 16, 12, 128, 3, 2, 1),
    ("1x1_s2_256_512", 2, 256, 16, 12, 512, 1, 2, 0),
    ("3x3_512_ragged_M", 1, 512, 8, 6, 512, 3, 1, 1),
    ("1x1_2048_512_big_K", 2, 2048, 8, 6, 512, 1, 1, 0),
    ("3x3_32_32_hrnet", 2, 32, 16, 12, 32, 3, 1, 1),
    ("1x1_bigM", 16, 64, 64, 48, 64, 1, 1, 0),
    ("3x3_bigM_128", 8, 128, 32, 24, 128, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_bn_relu_residual_vs_torch_cpu(case):
    name, B, Cin, H, W, Cout, k, s, p = case
    w = torch.from_numpy(synth.tensor_normal(1, name + "/w", (Cout, Cin, k, k), std=(2.0 / (Cin * k * k)) ** 0.5))
    x = torch.from_numpy(synth.tensor_normal(1, name + "/x", (B, Cin, H, W)))
    scale = torch.from_numpy(synth.tensor_uniform(1, name + "/s", (Cout,), 0.5, 1.5))
    shift = torch.from_numpy(synth.tensor_normal(1, name + "/b", (Cout,), std=0.3))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=s, padding=p)
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    res = torch.from_numpy(synth.tensor_normal(1, name + "/r", tuple(ref.shape)))
    ref_res = torch.relu(ref + res.double())

    y = _run_conv(x, lambda b, src: b.conv(src, w.to(DEV), stride=s, pad=p, scale=scale.to(DEV), shift=shift.to(DEV)))
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()
    assert err < 2e-6, err

    def with_res(b, src):
        b.p.shapes["res"] = tuple(ref.shape[2:]) + (Cout,)
        return b.conv(src, w.to(DEV), stride=s, pad=p, scale=scale.to(DEV), shift=shift.to(DEV), relu=True, res="res")

    B_, C_, H_, W_ = x.shape
    bld = engine.ProgramBuilder(H_, W_)
    bld.p.shapes["input"] = (H_, W_, C_)
    out = with_res(bld, "input")
    prog = bld.p
    prog.out_name = out
    prog.out_shape = prog.shapes[out]
    # inject the residual buffer by hand
    pool = prog._alloc(B, torch.device(DEV))
    pool["res"] = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = prog.run(x.permute(0, 2, 3, 1).contiguous().to(DEV)).cpu()
    err = (y.permute(0, 3, 1, 2).double() - ref_res).abs().max() / ref_res.abs().max()
    assert err < 2e-6, err


def test_stem_7x7_s2_on_nhwc4():
    x = torch.from_numpy(synth.input_images(2, seed=5, h=64, w=96))
    w = torch.from_numpy(synth.tensor_normal(2, "stem/w", (64, 3, 7, 7), std=(2.0 / 147) ** 0.5))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=3))
    b = engine.ProgramBuilder(64, 96)
    x4 = b.to_nhwc4("input")
    out = b.conv(x4, w.to(DEV), stride=2, pad=3, relu=True, name="conv1")
    pooled = b.maxpool(out)
    prog = b.p
    prog.out_name = pooled
    prog.out_shape = prog.shapes[pooled]
    y = prog.run(x.to(DEV)).cpu()
    refp = torch.nn.functional.max_pool2d(ref, 3, 2, 1)
    err = (y.permute(0, 3, 1, 2).double() - refp).abs().max() / refp.abs().max()
    assert err < 2e-6, err


@pytest.mark.parametrize("cin,cout,h,w,B", [(64, 256, 8, 6, 2), (2048, 256, 8, 6, 3), (256, 256, 32, 24, 4)])
def test_deconv_k4s2p1_phases_vs_torch_cpu(cin, cout, h, w, B):
    wt = torch.from_numpy(synth.tensor_normal(3, f"dc/{cin}", (cin, cout, 4, 4), std=(2.0 / (cin * 4)) ** 0.5))
    x = torch.from_numpy(synth.tensor_normal(3, f"dc/x{cin}", (B, cin, h, w)))
    scale = torch.from_numpy(synth.tensor_uniform(3, "dc/s", (cout,), 0.5, 1.5))
    shift = torch.from_numpy(synth.tensor_normal(3, "dc/b", (cout,), std=0.3))
    ref = torch.nn.functional.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1)
    ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    y = _run_conv(x, lambda b, src: b.deconv_k4s2p1(src, wt.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), relu=True))
    assert y.shape == (B, 2 * h, 2 * w, cout)
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()
    # exact-fp32 MFMA = k-ordered fmaf chain: error vs fp64 grows with K (3.5e-7 * sum|a*b| at K=4096, guide section 3)
    assert err < (1e-5 if cin * 4 >= 4096 else 2e-6), err


def test_duc_conv_with_fused_pixel_shuffle_and_final_nchw():
    B, cin, h, w = 2, 128, 16, 12
    wt = torch.from_numpy(synth.tensor_normal(4, "duc/w", (256, cin, 3, 3), std=(2.0 / (cin * 9)) ** 0.5))
    x = torch.from_numpy(synth.tensor_normal(4, "duc/x", (B, cin, h, w)))
    scale = torch.from_numpy(synth.tensor_uniform(4, "duc/s", (256,), 0.5, 1.5))
    shift = torch.from_numpy(synth.tensor_normal(4, "duc/b", (256,), std=0.3))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    ref = torch.nn.functional.pixel_shuffle(ref, 2)
    perm = TorchPacker.row_perm(256, "cpu")      # scale / shift go in the packed (sub-pixel-major) column order
    y = _run_conv(x, lambda b, src: b.conv(src, wt.to(DEV), pad=1, scale=scale[perm].to(DEV), shift=shift[perm].to(DEV), relu=True,
                                           pixel_shuffle=True))
    assert y.shape == (B, 2 * h, 2 * w, 64)
    assert (y.permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max() < 2e-6
    # bare pixel shuffle kernel
    xs = torch.from_numpy(synth.tensor_normal(4, "ps/x", (B, 64, 6, 4)))
    ys = _run_conv(xs, lambda b, src: b.pixel_shuffle(src))
    assert torch.equal(ys.permute(0, 3, 1, 2), torch.nn.functional.pixel_shuffle(xs, 2))
    # final layer: 3x3 + bias, 17 channels, NCHW store
    wf = torch.from_numpy(synth.tensor_normal(4, "fin/w", (17, 64, 3, 3), std=(2.0 / (64 * 9)) ** 0.5))
    bias = torch.from_numpy(synth.tensor_normal(4, "fin/b", (17,), std=0.1))
    xf = torch.from_numpy(synth.tensor_normal(4, "fin/x", (B, 64, 16, 12)))
    reff = torch.nn.functional.conv2d(xf.double(), wf.double(), bias.double(), padding=1)
    bld = engine.ProgramBuilder(16, 12)
    bld.p.shapes["input"] = (16, 12, 64)
    bld.conv("input", wf.to(DEV), pad=1, shift=bias.to(DEV), out_nchw=True, dst="heat")
    bld.p.out_shape = (17, 16, 12)
    yf = bld.p.run(xf.permute(0, 2, 3, 1).contiguous().to(DEV)).cpu()
    assert (yf.double() - reff).abs().max() / reff.abs().max() < 2e-6




def test_c_abi_packing_matches_the_layout_spec_bit_for_bit():
    """sp_pack_conv_weights / sp_pack_deconv_k4s2p1 / sp_fold_bn (csrc/pack.hip, what engine.HipPacker calls) against the torch
    restatement of the layouts (tests/desc_interp.TorchPacker): plain, padded-stem, x-paired bf16 stem, PixelShuffle row order,
    transposed-conv phases, fp32 and bf16 destinations - identical bits, padding included."""
    hp, tp = engine.HipPacker(), TorchPacker()
    cases = [((64, 32, 3, 3), {}), ((256, 64, 1, 1), {}), ((17, 256, 1, 1), {}), ((17, 128, 3, 3), {}),
             ((64, 3, 7, 7), dict(c_in_pad=4, taps_w_pad=8)), ((64, 3, 3, 3), dict(c_in_pad=4, taps_w_pad=4)),
             ((64, 3, 7, 7), dict(c_in_pad=8, taps_w_pad=4, pair_s0=1, bf16=True)), ((64, 3, 3, 3), dict(c_in_pad=8, taps_w_pad=2, pair_s0=1, bf16=True)),
             ((64, 3, 5, 5), dict(c_in_pad=8, taps_w_pad=3, pair_s0=0, bf16=True)),
             ((1024, 512, 3, 3), dict(pixel_shuffle=True)), ((512, 256, 3, 3), dict(pixel_shuffle=True, bf16=True)), ((96, 40, 3, 3), dict(bf16=False))]
    for i, (shape, kw) in enumerate(cases):
        w = torch.from_numpy(synth.tensor_normal(21, f"pack/{i}", shape))
        for bf in ({kw.get("bf16", False)} | ({True} if "pair_s0" not in kw and shape[1] % 8 == 0 else set())):
            k2 = dict(kw, bf16=bf)
            got, want = hp.conv(w.to(DEV), **k2), tp.conv(w, **k2)
            assert got[1:] == want[1:], (shape, k2, got[1:], want[1:])
            assert got[0].dtype == want[0].dtype and torch.equal(got[0].cpu(), want[0]), (shape, k2)
    for cin, cout in ((64, 256), (2048, 256), (256, 17)):
        w = torch.from_numpy(synth.tensor_normal(21, f"packd/{cin}", (cin, cout, 4, 4)))
        for bf in (False, True):
            got, want = hp.deconv(w.to(DEV), bf16=bf), tp.deconv(w, bf16=bf)
            assert got[1] == want[1] and torch.equal(got[0].cpu(), want[0]), (cin, cout, bf)
    for C, ps in ((64, False), (1024, True), (17, False)):
        g, b_ = (torch.from_numpy(synth.tensor_uniform(21, f"bn/{C}/{n}", (C,), 0.5, 1.5)) for n in "gb")
        m = torch.from_numpy(synth.tensor_normal(21, f"bn/{C}/m", (C,), std=0.3))
        v = torch.from_numpy(synth.tensor_uniform(21, f"bn/{C}/v", (C,), 1e-3, 4.0))
        got, want = hp.fold_bn(g.to(DEV), b_.to(DEV), m.to(DEV), v.to(DEV), pixel_shuffle=ps), tp.fold_bn(g, b_, m, v, pixel_shuffle=ps)
        assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1]), C


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("O,groups,panel", [(128, 32, 64), (256, 32, 64), (1024, 32, 64), (2048, 32, 64), (256, 4, 64), (256, 2, 128)])
def test_grouped_pack_and_conv_vs_torch(O, groups, panel, bf16):
    """Grouped 3x3 convolution (ResNeXt's conv2, nets/pose_resnet_dconv.py:101): sp_pack_conv_weights_grouped equals the layout restated in
    tests/desc_interp.TorchPacker bit for bit, and the launch (sp_conv_desc.c_in_group = tile_n = the panel) equals torch's grouped conv2d in
    float64 on the same operands, on every tile_m the kernel offers for that panel, stride 1 and 2."""
    lib = _lib.lib()
    cpg = O // groups
    g = torch.Generator().manual_seed(O + groups)
    w = (torch.randn(O, cpg, 3, 3, generator=g) * (2.0 / (cpg * 9)) ** 0.5)
    if bf16:
        w = w.bfloat16().float()
    packed = engine.HipPacker().grouped(w.to(DEV), groups, panel, bf16=bf16)
    torch.cuda.synchronize()
    assert torch.equal(packed.cpu(), TorchPacker().grouped(w, groups, panel, bf16=bf16))
    for stride, (B, H, W) in ((1, (2, 9, 7)), (2, (3, 12, 10))):
        x = torch.randn(B, O, H, W, generator=g)
        if bf16:
            x = x.bfloat16().float()
        ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=1, groups=groups)
        b = engine.ProgramBuilder(H, W, dtype="bf16" if bf16 else "fp32")
        b.p.shapes["input"] = (H, W, O)
        out = b.conv("input", w.to(DEV), stride=stride, pad=1, name="c", groups=groups)
        prog = b.p
        op = [o for o in prog.ops if o.kind == "conv"][0]
        assert op.desc.c_in_group == panel and not op.direct
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16 if bf16 else torch.float32)
        op.desc.batch = B
        res = []
        for cand in prog._candidates(lib, op):
            op.desc.tile_m, op.desc.tile_n, op.desc.kernel = cand
            y = torch.full((B,) + tuple(prog.shapes[out]), float("nan"), dtype=xin.dtype, device=DEV)
            _lib.check(lib.sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), None, None, None, _lib.ptr(y), _lib.current_stream()), str(cand))
            torch.cuda.synchronize()
            res.append(y.float().cpu())
        assert len(res) >= 2
        for r in res[1:]:
            assert torch.equal(r, res[0])                       # every tile the same bits
        got = res[0].permute(0, 3, 1, 2).double()
        assert not torch.isnan(got).any()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < (6e-3 if bf16 else 3e-6), (stride, err)


def test_ctypes_only_pack_and_run_one_conv():
    """What a maintainer binding only include/simple_pose_hip.h does: no simple_pose_amd.engine, just the
    C entry points - sp_conv_packed_dims + sp_pack_conv_weights + sp_fold_bn + a hand-filled sp_conv_desc + sp_conv2d_fwd -
    reproduce conv3x3 + eval-mode BatchNorm + ReLU of the reference block (pose_resnet_dconv.py:112-120) in fp64."""
    import ctypes
    lib = _lib.lib()
    B, Cin, H, W, Cout = 3, 64, 16, 12, 128
    w = torch.from_numpy(synth.tensor_normal(22, "c/w", (Cout, Cin, 3, 3), std=(2.0 / (Cin * 9)) ** 0.5))
    x = torch.from_numpy(synth.tensor_normal(22, "c/x", (B, Cin, H, W)))
    g, b_ = (torch.from_numpy(synth.tensor_uniform(22, f"c/{n}", (Cout,), 0.5, 1.5)) for n in "gb")
    m = torch.from_numpy(synth.tensor_normal(22, "c/m", (Cout,), std=0.3))
    v = torch.from_numpy(synth.tensor_uniform(22, "c/v", (Cout,), 0.2, 2.0))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    ref = torch.relu(torch.nn.functional.batch_norm(ref, m.double(), v.double(), g.double(), b_.double(), False, 0.0, 1e-5))
    n_pad, k_pad = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.sp_conv_packed_dims(Cout, 9 * Cin, 0, ctypes.byref(n_pad), ctypes.byref(k_pad)) == 0
    st = _lib.current_stream()
    dw = w.to(DEV)
    packed = torch.empty((n_pad.value, k_pad.value), dtype=torch.float32, device=DEV)
    _lib.check(lib.sp_pack_conv_weights(_lib.ptr(dw), Cout, Cin, 3, 3, Cin, 3, 0, -1, n_pad.value, k_pad.value, _lib.ptr(packed), 0, st))
    scale, shift = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    dg, db, dm, dv = (t.to(DEV) for t in (g, b_, m, v))
    _lib.check(lib.sp_fold_bn(_lib.ptr(dg), _lib.ptr(db), _lib.ptr(dm), _lib.ptr(dv), Cout, 1e-5, 0, _lib.ptr(scale), _lib.ptr(shift), st))
    d = _lib.ConvDesc()
    d.batch, d.in_h, d.in_w, d.c_in = B, H, W, Cin
    d.grid_h, d.grid_w, d.c_out, d.n_pad = H, W, Cout, n_pad.value
    d.taps_h, d.taps_w, d.k_pad, d.stride = 3, 3, k_pad.value, 1
    d.dy0, d.dy_step, d.dx0, d.dx_step = -1, 1, -1, 1
    d.out_h, d.out_w, d.out_c = H, W, Cout
    d.oy_mul = d.ox_mul = d.phases_y = d.phases_x = 1
    d.flags = _lib.SP_CONV_RELU
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=DEV)
    _lib.check(lib.sp_conv2d_fwd(d, _lib.ptr(xn), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift), None, _lib.ptr(y), st))
    torch.cuda.synchronize()
    err = (y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()
    assert err < 2e-6, err
    # argument validation up front: a destination too small for the padded matrix is refused with a message, nothing is launched
    assert lib.sp_pack_conv_weights(_lib.ptr(dw), Cout, Cin, 3, 3, Cin, 3, 0, -1, n_pad.value, 9 * Cin - 32, _lib.ptr(packed), 0, st) == -1
    assert b"k_pad" in lib.sp_last_error()


# ---------------------------------------------------------------------------------------------- networks
def _load(mod, head, seed):
    m = mod.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(DEV).eval()


# pinned from gpurun_out/measured_parity.json (the `measured` fixture): heat-map error this kernel showed, and the end-to-end
# fractions (identical arg-max cells, joints within 1e-3 px) minus one percentage point
# (round 2: heat maps 2.59e-6 / 2.22e-6; identical cells 100 % / 100 %; within 1e-3 px 94.1 % / 100 %, worst joint 0.019 / 0.0007 px)
FWD_REL_MEASURED = {"dconv": 2.6e-6, "duc": 2.3e-6}
E2E_BARS = {"dconv": (0.99, 0.93), "duc": (0.99, 0.99)}


@pytest.mark.parametrize("mod,head,fname", [(pose_resnet_dconv, "dconv", "g1_dconv_fwd.npz"),
                                            (pose_resnet_duc, "duc", "g2_duc_fwd.npz")])
def test_forward_vs_reference_golden(golden, measured, mod, head, fname):
    g = golden(fname)
    m = _load(mod, head, int(g["seed"]))
    x = _cuda(synth.input_images(int(g["batch"]), int(g["seed"])))
    with torch.no_grad():
        hm = m(x)
    assert hm.shape == (2, 17, 64, 48) and hm.dtype == torch.float32 and hm.is_cuda
    ref = g["heat_maps"]
    rel = np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max()
    measured("heat_map_rel_err", rel, 1e-4)
    assert rel <= 1e-4, rel                       # BASELINE.json: heat maps within 1e-4 rel fp32
    assert rel <= 3 * FWD_REL_MEASURED[head], rel  # and no more than 3x what this kernel measured when the bar was pinned
    # end to end: decoded key points.  Noise-like maps make -H^-1 g ill-conditioned, so the
    # contract is: arg-max cell identical, and the bulk of joints within 1e-3 px; the tail is reported, not hidden.
    tinv = synth.trans_inv_batch(2)
    ref_kps, _ = pose_oracle.decode_gauss_taylor(ref, tinv)
    kps, _ = GaussTaylorKeyPointDecoder()(hm, _cuda(tinv))
    co, _ = BasicKeyPointDecoder.heat_map_to_axis(hm)
    rco, _ = pose_oracle.heat_map_to_axis(ref)
    same_cell = (co.cpu().numpy() == rco).all(-1).mean()
    err = np.abs(kps.cpu().numpy() - ref_kps).max(-1) / 4.0      # heat-map px
    within = (err <= 1e-3).mean()
    measured("argmax_cell_match_fraction", same_cell, E2E_BARS[head][0])
    measured("joints_within_1e-3px_fraction", within, E2E_BARS[head][1])
    measured("worst_joint_px", err.max())
    assert same_cell >= E2E_BARS[head][0], same_cell
    assert within >= E2E_BARS[head][1], (within, err.max())


WIDE_NETS = [("dconv", 8), ("duc", 8), ("hrnet_w32", 4), ("dconv_se", 4)]


@pytest.mark.parametrize("tag,B", WIDE_NETS, ids=[t for t, _ in WIDE_NETS])
def test_forward_vs_reference_on_the_wide_set(golden, measured, tag, B):
    """Round-3 verdict, weak 1: G1-G3 hold one or two reference images.  g10_fwd_wide.npz = 8 / 8 / 4 / 4 DISTINCT images through the real
    reference (nets/pose_resnet_dconv.py, pose_resnet_duc.py, pose_hrnet.py, SELayer variant) with a second set of conditioned weights: the
    HIP forward matches the sub-sampled maps (every 4th row / column) within the 1e-4 contract, the per-joint sums / norms / maxima, the
    arg-max cells, and the decoded key points; one batch of all images = each image alone (bitwise)."""
    import os
    g = golden("g10_fwd_wide.npz")
    seed = int(g["w_seed"])
    if tag == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), seed)
    elif tag == "dconv_se":
        m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
        sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv", se=True), seed)
    else:
        m = (pose_resnet_dconv if tag == "dconv" else pose_resnet_duc).resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(tag), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    x = synth.input_images(B, int(g["x_seed"]))
    with torch.no_grad():
        hm = m(_cuda(x))
        one = m(_cuda(x[B - 1:B]))
    assert torch.equal(one[0], hm[B - 1])
    h = hm.cpu().numpy()
    scale = np.abs(g[f"{tag}/heat_max"]).max()
    rel = np.abs(h[:, :, ::4, ::4] - g[f"{tag}/heat_sub"]).max() / scale
    measured("heat_sub_rel_err", rel, 1e-4)
    assert rel <= 1e-4, rel
    flat = h.reshape(B, 17, -1)
    e_l2 = np.abs(np.sqrt((flat.astype(np.float64) ** 2).sum(-1)) - g[f"{tag}/heat_l2"]).max() / g[f"{tag}/heat_l2"].max()
    e_sum = np.abs(flat.astype(np.float64).sum(-1) - g[f"{tag}/heat_sum"]).max() / (scale * flat.shape[-1] ** 0.5)
    measured("heat_l2_rel_err", e_l2, 1e-5)
    measured("heat_sum_err_over_scale_sqrt_n", e_sum, 1e-4)
    assert e_l2 <= 1e-5 and e_sum <= 1e-4
    assert np.abs(flat.max(-1) - g[f"{tag}/heat_max"]).max() <= 1e-4 * scale
    same = (flat.argmax(-1) == g[f"{tag}/heat_argmax"]).mean()
    kps, mv = GaussTaylorKeyPointDecoder()(hm, _cuda(synth.trans_inv_batch(B)))
    err = np.abs(kps.cpu().numpy() - g[f"{tag}/gt_kps"]).max(-1) / 4.0
    within = (err <= 1e-3).mean()
    measured("argmax_cell_match_fraction", same, 0.99)
    measured("joints_within_1e-3px_fraction", within, 0.9)
    measured("worst_joint_px", err.max())
    assert same >= 0.99 and within >= 0.9, (same, within)


RESNET_VARIANTS = [("resnet18", "dconv", False), ("resnet34", "duc", False), ("wide_resnet50_2", "dconv", False), ("resnet18", "dconv", True),
                   ("resnext50_32x4d", "dconv", False), ("resnext101_32x8d", "duc", False)]      # (round 5: the grouped resnext factories, golden g12 from the real reference)


@pytest.mark.parametrize("arch,head,se", RESNET_VARIANTS, ids=[f"{a}_{h}" + ("_se" if s else "") for a, h, s in RESNET_VARIANTS])
def test_resnet_variants_forward_vs_reference_golden(golden, measured, arch, head, se):
    """Round-3 verdict, missing 5: the reference's other factories (nets/pose_resnet_dconv.py:282-403, pose_resnet_duc.py): resnet18 / resnet34
    (BasicBlock: two 3x3 convs, projection shortcut only where the shape changes, head from 512 channels) and wide_resnet50_2 (Bottleneck with
    twice the inner width), one with SELayers - HIP forward against g11 (real reference, 2 images) within the 1e-4 contract; bf16 operands
    within the bf16 bar of the fp32 program."""
    g = golden("g12_resnext.npz" if arch.startswith("resnext") else "g11_resnet_variants.npz")
    tag = f"{arch}_{head}" + ("_se" if se else "")
    m = getattr(pose_resnet_dconv if head == "dconv" else pose_resnet_duc, arch)(pretrained=False, num_classes=17, reduction=se)
    layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
    sd = synth.conditioned_state_dict(layout, int(g["w_seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    x = _cuda(synth.input_images(2, int(g["x_seed"])))
    with torch.no_grad():
        hm = m(x)
        m.compute_dtype = "bf16"
        hb = m(x)
    h = hm.cpu().numpy()
    scale = np.abs(g[f"{tag}/heat_max"]).max()
    rel = np.abs(h[:, :, ::4, ::4] - g[f"{tag}/heat_sub"]).max() / scale
    measured("heat_sub_rel_err", rel, 1e-4)
    assert rel <= 1e-4, rel
    flat = h.reshape(2, 17, -1)
    e_l2 = np.abs(np.sqrt((flat.astype(np.float64) ** 2).sum(-1)) - g[f"{tag}/heat_l2"]).max() / g[f"{tag}/heat_l2"].max()
    assert e_l2 <= 1e-5, e_l2
    same = (flat.argmax(-1) == g[f"{tag}/heat_argmax"]).mean()
    kps, _ = GaussTaylorKeyPointDecoder()(hm, _cuda(synth.trans_inv_batch(2)))
    within = (np.abs(kps.cpu().numpy() - g[f"{tag}/gt_kps"]).max(-1) / 4.0 <= 1e-3).mean()
    measured("argmax_cell_match_fraction", same, 0.97)
    measured("joints_within_1e-3px_fraction", within, 0.85)
    assert same >= 0.97 and within >= 0.85, (same, within)
    relb = float((hb - hm).abs().max() / hm.abs().max())
    measured("bf16_vs_fp32_rel", relb, 3e-2)
    assert 1e-5 < relb <= 3e-2, relb


@pytest.mark.parametrize("head,H,W,dtype", [("dconv", 384, 288, "fp32"), ("duc", 128, 96, "fp32"), ("dconv", 320, 224, "bf16"), ("duc", 384, 288, "bf16")])
def test_resnets_at_other_resolutions_vs_oracle(measured, head, H, W, dtype):
    """The ResNets at input sizes other than the 256x192 of the golden vectors (the reference's 384x288 setting; small and odd-tile
    sizes): every launch descriptor, tile count and phase geometry changes with the resolution.  Checked against the forward oracle
    (torch-CPU restatement, itself pinned to the goldens at 256x192): fp32 within the BASELINE 1e-4, bf16 within its bar."""
    mod = pose_resnet_dconv if head == "dconv" else pose_resnet_duc
    m = _load(mod, head, 4)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), 4).items()}
    x = synth.input_images(3, 12, h=H, w=W)
    with torch.no_grad():
        ref = nets_oracle.FORWARDS["resnet50_" + head](sd, torch.from_numpy(x)).numpy()
        m.compute_dtype = dtype if dtype == "bf16" else "fp32"
        hm = m(_cuda(x))
    assert hm.shape == (3, 17, H // 4, W // 4) and hm.dtype == torch.float32
    rel = np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max()
    bar = 8e-6 if dtype == "fp32" else 2e-2        # measured (round 2): 2.6e-6 / 1.8e-6 fp32, 1.11e-2 / 0.95e-2 bf16 (BASELINE contract: 1e-4 fp32)
    measured("heat_map_rel_err", rel, bar)
    assert rel <= bar, rel


@pytest.mark.parametrize("factory,head", [("resnet101", "dconv"), ("resnet152", "duc")])
def test_deeper_resnet_factories_vs_oracle(measured, factory, head):
    """resnet101 / resnet152 (nets/pose_resnet_dconv.py:318-339 and the DUC twin): same bottleneck trunk at depths [3,4,23,3] / [3,8,36,3];
    eval forward against the torch-CPU forward oracle on the model's own state_dict layout."""
    from simple_pose_amd.nets import pose_resnet_duc
    mod = pose_resnet_dconv if head == "dconv" else pose_resnet_duc
    m = getattr(mod, factory)(pretrained=False, num_classes=17)
    layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
    sd = synth.conditioned_state_dict(layout, seed=3)
    assert nets_oracle.blocks_of(sd) == {"resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}[factory]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    x = synth.input_images(2, 9, h=128, w=96)
    with torch.no_grad():
        got = m(torch.from_numpy(x).to(DEV)).cpu().numpy()
        ref = nets_oracle.FORWARDS["resnet50_" + head]({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(x)).numpy()
    err = np.abs(got - ref).max() / np.sqrt((ref * ref).mean())
    measured("heat_rel_err", err, 2e-5)
    assert got.shape == (2, 17, 32, 24) and err < 2e-5


@pytest.mark.parametrize("head,B,H,W", [("dconv", 3, 256, 192), ("duc", 2, 128, 96), ("dconv", 1, 96, 160)])
def test_fused_bottlenecks_equal_the_per_conv_program_bitwise(head, B, H, W):
    """model.fuse_bottlenecks (default on): layer1.1 / layer1.2 (identity shortcut, 256 -> 64 -> 64 -> 256) as ONE launch each (sp_bottleneck_c64) and
    - round 6 - layer1.0's conv3 + projection shortcut as one launch (sp_dual_pw_bf16: the 256-channel shortcut tensor is never written) give
    the heat maps of the conv-by-conv bf16 program bit for bit - same accumulation chains, intermediates rounded to bf16 at the same
    places; incl. sizes whose 16x8 tiles are ragged (96x160 input: 24x40 maps) and row counts that are no multiple of the 128-pixel tile."""
    m = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[head].resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), 6)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = "bf16"
    m.autotune = False
    x = _cuda(synth.input_images(B, 21, h=H, w=W))
    with torch.no_grad():
        m.fuse_bottlenecks = False
        ref = m(x).clone()
        n_ref = len(m.hip_program(x).ops)
        assert not any(op.kind == "bneck64" for op in m.hip_program(x).ops)
        m.fuse_bottlenecks = True                  # (the default)
        got = m(x)
        prog = m.hip_program(x)
    assert sum(op.kind == "bneck64" for op in prog.ops) == 2 and sum(op.kind == "dual1x1" for op in prog.ops) == 1 and len(prog.ops) == n_ref - 5
    assert torch.equal(got, ref)


@pytest.mark.parametrize("rows,relu", [(64 * 9, True), (1000, True), (50, False), (64 * 700 + 3, True)])
def test_dual_pointwise_tail_fp32_equals_the_two_launches_bitwise(rows, relu):
    """sp_dual_pw_f32 (csrc/conv_pw.hip: the fp32 twin, on the headline configuration's layer1.0) against the two fp32 launches it replaces, bit for bit -
    the shortcut's tensor is an fp32 tensor there, so nothing is rounded differently - and against float64."""
    lib = _lib.lib()
    t = torch.from_numpy(synth.tensor_normal(9, "dual32/t", (rows, 64)))
    x = torch.from_numpy(synth.tensor_normal(9, "dual32/x", (rows, 64)))
    w3 = torch.from_numpy(synth.tensor_normal(9, "dual32/w3", (256, 64, 1, 1), std=0.2))
    wd = torch.from_numpy(synth.tensor_normal(9, "dual32/wd", (256, 64, 1, 1), std=0.2))
    s3, h3, sd_, hd = (torch.from_numpy(synth.tensor_uniform(9, "dual32/" + n, (256,), lo, hi)).float().to(DEV)
                       for n, lo, hi in (("s3", 0.5, 1.5), ("h3", -0.3, 0.3), ("sd", 0.5, 1.5), ("hd", -0.3, 0.3)))
    b = engine.ProgramBuilder(1, rows, dtype="fp32")
    b.p.shapes["t"] = (1, rows, 64)
    b.p.shapes["x"] = (1, rows, 64)
    b.fuse_tail = False
    r = b.conv("x", wd.to(DEV), scale=sd_, shift=hd, name="ds")
    b.conv("t", w3.to(DEV), scale=s3, shift=h3, relu=relu, res=r, name="c3")
    ops = {o.name: o for o in b.p.ops}
    tg, xg = t.to(DEV).view(1, 1, rows, 64), x.to(DEV).view(1, 1, rows, 64)
    rbuf = torch.empty((1, 1, rows, 256), dtype=torch.float32, device=DEV)
    two = torch.full((1, 1, rows, 256), float("nan"), dtype=torch.float32, device=DEV)
    for o, src, res, dst in ((ops["ds"], xg, None, rbuf), (ops["c3"], tg, rbuf, two)):
        o.desc.batch = 1
        _lib.check(lib.sp_conv2d_fwd(o.desc, _lib.ptr(src), _lib.ptr(o.w), _lib.ptr(o.scale), _lib.ptr(o.shift), _lib.ptr(res), _lib.ptr(dst),
                                     _lib.current_stream()), o.name)
    one = torch.full((1, 1, rows, 256), float("nan"), dtype=torch.float32, device=DEV)
    assert lib.sp_dual_pw_f32_ok(rows, 64, 64, 256) == 1
    _lib.check(lib.sp_dual_pw_f32(_lib.ptr(tg), _lib.ptr(ops["c3"].w), _lib.ptr(s3), _lib.ptr(h3), _lib.ptr(xg), _lib.ptr(ops["ds"].w), _lib.ptr(sd_), _lib.ptr(hd),
                                  _lib.ptr(one), rows, 64, 64, 256, int(relu), _lib.current_stream()), "dual f32")
    torch.cuda.synchronize()
    assert not torch.isnan(one).any()
    assert torch.equal(one.view(torch.int32), two.view(torch.int32)), int((one.view(torch.int32) != two.view(torch.int32)).sum())
    ref = ((t.double() @ w3.double().view(256, 64).T) * s3.cpu().double() + h3.cpu().double() +
           (x.double() @ wd.double().view(256, 64).T) * sd_.cpu().double() + hd.cpu().double())
    if relu:
        ref = torch.relu(ref)
    assert float((one.cpu().double().view(rows, 256) - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("rows,relu", [(128 * 7, True), (1000, True), (77, False), (128 * 300 + 5, True)])
def test_dual_pointwise_tail_equals_the_two_launches_bitwise(rows, relu):
    """sp_dual_pw_bf16 alone: y = [relu](bn3(t . W3^T) + bn_d(x . Wd^T)) (nets/pose_resnet_dconv.py:99-103,120-131) against the two launches it
    replaces - the shortcut's 1x1 conv (bf16 store) and conv3 with that tensor as its residual - bit for bit, on row counts below, at and far
    above one 128-pixel tile (several tiles per workgroup, a ragged last tile), and against float64 on the same bf16 operands."""
    lib = _lib.lib()
    t = torch.from_numpy(synth.tensor_normal(9, "dual/t", (rows, 64))).bfloat16()
    x = torch.from_numpy(synth.tensor_normal(9, "dual/x", (rows, 64))).bfloat16()
    w3 = torch.from_numpy(synth.tensor_normal(9, "dual/w3", (256, 64, 1, 1), std=0.2)).bfloat16().float()
    wd = torch.from_numpy(synth.tensor_normal(9, "dual/wd", (256, 64, 1, 1), std=0.2)).bfloat16().float()
    s3, h3, sd_, hd = (torch.from_numpy(synth.tensor_uniform(9, "dual/" + n, (256,), lo, hi)).float().to(DEV)
                       for n, lo, hi in (("s3", 0.5, 1.5), ("h3", -0.3, 0.3), ("sd", 0.5, 1.5), ("hd", -0.3, 0.3)))
    b = engine.ProgramBuilder(1, rows, dtype="bf16")                     # a [1 x rows] "image" of 64 channels
    b.p.shapes["t"] = (1, rows, 64)
    b.p.shapes["x"] = (1, rows, 64)
    b.fuse_tail = False
    r = b.conv("x", wd.to(DEV), scale=sd_, shift=hd, name="ds")
    y2 = b.conv("t", w3.to(DEV), scale=s3, shift=h3, relu=relu, res=r, name="c3")
    ops = {o.name: o for o in b.p.ops}
    tg, xg = t.to(DEV).view(1, 1, rows, 64), x.to(DEV).view(1, 1, rows, 64)
    rbuf = torch.empty((1, 1, rows, 256), dtype=torch.bfloat16, device=DEV)
    two = torch.full((1, 1, rows, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    for o, src, res, dst in ((ops["ds"], xg, None, rbuf), (ops["c3"], tg, rbuf, two)):
        o.desc.batch = 1
        _lib.check(lib.sp_conv2d_fwd(o.desc, _lib.ptr(src), _lib.ptr(o.w), _lib.ptr(o.scale), _lib.ptr(o.shift), _lib.ptr(res), _lib.ptr(dst),
                                     _lib.current_stream()), o.name)
    one = torch.full((1, 1, rows, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert lib.sp_dual_pw_bf16_ok(rows, 64, 64, 256) == 1
    _lib.check(lib.sp_dual_pw_bf16(_lib.ptr(tg), _lib.ptr(ops["c3"].w), _lib.ptr(s3), _lib.ptr(h3), _lib.ptr(xg), _lib.ptr(ops["ds"].w), _lib.ptr(sd_), _lib.ptr(hd),
                                   _lib.ptr(one), rows, 64, 64, 256, int(relu), _lib.current_stream()), "dual")
    torch.cuda.synchronize()
    assert not torch.isnan(one.float()).any()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16)), int((one.view(torch.int16) != two.view(torch.int16)).sum())
    rd = (x.double() @ wd.double().view(256, 64).T) * sd_.cpu().double() + hd.cpu().double()
    ref = (t.double() @ w3.double().view(256, 64).T) * s3.cpu().double() + h3.cpu().double() + rd.bfloat16().double()
    if relu:
        ref = torch.relu(ref)
    assert float((one.float().cpu().double().view(rows, 256) - ref).abs().max() / ref.abs().max()) < 8e-3


@pytest.mark.parametrize("head,B,H,W", [("dconv", 3, 256, 192), ("duc", 1, 96, 160)])
def test_fp32_dual_tail_equals_the_per_conv_program_bitwise(head, B, H, W):
    """fp32 (the headline configuration's arithmetic): layer1.0's conv3 + projection shortcut as one launch (sp_dual_pw_f32, round 6) gives the heat
    maps of the conv-by-conv program bit for bit (`fuse_bottlenecks` switches it; the fused identity Bottlenecks exist in bf16 only)."""
    m = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[head].resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), 6)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.autotune = False
    x = _cuda(synth.input_images(B, 23, h=H, w=W))
    with torch.no_grad():
        m.fuse_bottlenecks = False
        ref = m(x).clone()
        n_ref = len(m.hip_program(x).ops)
        m.fuse_bottlenecks = True
        got = m(x)
        prog = m.hip_program(x)
    assert sum(op.kind == "dual1x1" for op in prog.ops) == 1 and not any(op.kind == "bneck64" for op in prog.ops) and len(prog.ops) == n_ref - 1
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("head,B,H,W", [("dconv", 3, 256, 192), ("duc", 2, 128, 96), ("dconv", 1, 96, 160), ("dconv", 5, 64, 64)])
def test_fused_stem_equals_the_three_launch_stem_bitwise(head, B, H, W, dtype):
    """model.fuse_stem (the default): conv1 + bn1 + relu + maxpool (pose_resnet_dconv.py:158-162) as ONE launch on the fp32 NCHW image
    (sp_stem7_pool) gives the heat maps of the layout-change / implicit-GEMM / pooling program bit for bit in both compute dtypes -
    the same MFMA instruction on the same k positions in the same order, only all-padding instructions dropped."""
    m = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[head].resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), 8)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = dtype
    m.autotune = False
    x = _cuda(synth.input_images(B, 23, h=H, w=W))
    with torch.no_grad():
        m.fuse_stem = False
        ref = m(x).clone()
        n_ref = len(m.hip_program(x).ops)
        m.fuse_stem = True
        got = m(x)
        prog = m.hip_program(x)
    assert prog.ops[0].kind == "stem7" and len(prog.ops) == n_ref - 2
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,H,W", [(2, 256, 192), (3, 64, 96), (1, 36, 52), (2, 4, 4), (1, 130, 34)])
def test_fused_hrnet_stem_equals_the_three_launches_bitwise(B, H, W):
    """sp_hrnet_stem (bf16: conv1 3x3 s2 + bn1 + relu + conv2 3x3 s2 + bn2 + relu, pose_hrnet.py:419-425, one launch on the fp32 NCHW image)
    against sp_nchw_to_nhwc4_bf16 -> sp_conv2d_fwd -> sp_conv2d_fwd: bit for bit, incl. sizes whose 8 x 8 output tiles are ragged and
    images smaller than a tile; conv2's zero padding is conv1 positions OUTSIDE conv1's output, not conv1 of a padded image."""
    g = torch.Generator().manual_seed(H * 1000 + W + 7)
    w1 = torch.randn((64, 3, 3, 3), generator=g).to(DEV) * 0.2
    w2 = torch.randn((64, 64, 3, 3), generator=g).to(DEV) * 0.05
    s1, h1 = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
    s2, h2 = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
    x = torch.randn((B, 3, H, W), generator=g).to(DEV)
    outs = []
    for fuse in (False, True):
        b = engine.ProgramBuilder(H, W, "bf16")
        b.fuse_stem = fuse
        out = b.hrnet_stem("input", w1, s1, h1, w2, s2, h2)
        prog = b.p
        assert [op.kind for op in prog.ops] == (["hstem"] if fuse else ["to_nhwc4", "conv", "conv"])
        bufs = dict(prog._alloc(B, x.device))
        bufs["input"] = x
        for op in prog.ops:
            prog._launch(_lib.lib(), op, bufs, B, _lib.current_stream())
        torch.cuda.synchronize()
        outs.append(bufs[out].clone())
    assert outs[0].numel() == B * ((H // 2 + 1) // 2) * ((W // 2 + 1) // 2) * 64
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert float(outs[1].float().abs().max()) > 0


def test_hrnet_with_fused_stem_equals_the_per_conv_program_bitwise():
    """model.fuse_stem on PoseHighResolutionNet (bf16, the default): same heat maps as the per-conv program, bit for bit; fp32 keeps the
    per-conv stem; uint8 crops run the op's three-launch definition."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    x = _cuda(synth.input_images(3, 5, h=128, w=96))
    with torch.no_grad():
        assert not any(op.kind == "hstem" for op in net.hip_program(x).ops)          # fp32
        net.compute_dtype = "bf16"
        net.fuse_stem = False
        ref = net(x).clone()
        net.fuse_stem = True
        got = net(x)
        assert net.hip_program(x).ops[0].kind == "hstem"
        assert torch.equal(got, ref)
        crops = torch.randint(0, 256, (2, 128, 96, 3), dtype=torch.uint8, device=DEV)
        a = net.forward_crops(crops).clone()
        net.fuse_stem = False
        assert torch.equal(net.forward_crops(crops), a)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,H,W", [(2, 70, 50), (1, 8, 8), (3, 34, 130), (2, 256, 192)])
def test_fused_stem_on_uint8_crops_bitwise(B, H, W, dtype):
    """sp_stem7_pool_u8: uint8 BGR crops [B,H,W,3] normalised (datasets/coco.py:136) while the patch is loaded, against
    sp_u8hwc_bgr_to_nhwc -> sp_conv2d_fwd -> sp_maxpool3x3s2_nhwc on the same crops - bit for bit, incl. black pixels next to the zero
    padding (a byte 0 normalises to -mean, padding stays 0) and sizes with ragged tiles."""
    g = torch.Generator().manual_seed(H * 1000 + W + 1)
    w = torch.randn((64, 3, 7, 7), generator=g).to(DEV) * 0.1
    scale = (torch.rand(64, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(64, generator=g) * 0.3).to(DEV)
    crops = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    crops[:, :3, :, :] = 0                                   # black rows at the border
    crops = crops.to(DEV)
    outs = []
    for fuse in (False, True):
        b = engine.ProgramBuilder(H, W, dtype)
        b.fuse_stem = fuse
        out = b.stem_pool("input", w, scale, shift)
        prog = b.p
        bufs = dict(prog._alloc(B, crops.device))
        bufs["input"] = crops
        for op in prog.ops:
            prog._launch(_lib.lib(), op, bufs, B, _lib.current_stream())
        torch.cuda.synchronize()
        outs.append(bufs[out].clone())
    view = torch.int16 if dtype == "bf16" else torch.int32
    assert torch.equal(outs[0].view(view), outs[1].view(view))
    assert float(outs[1].float().abs().max()) > 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,H,W", [(2, 70, 50), (1, 8, 8), (3, 34, 130), (1, 258, 62)])
def test_fused_stem_kernel_on_ragged_sizes_bitwise(B, H, W, dtype):
    """sp_stem7_pool through the C ABI against sp_nchw_to_nhwc4 -> sp_conv2d_fwd -> sp_maxpool3x3s2_nhwc on sizes whose pooled maps are
    not multiples of the 8 x 8 tile (the nets only see multiples of 32): ragged tiles, image borders inside every patch, one-tile
    images.  Values include negatives after BatchNorm (ReLU clamps) and a NaN and an inf pixel (treated alike by both paths)."""
    g = torch.Generator().manual_seed(H * 1000 + W)
    w = torch.randn((64, 3, 7, 7), generator=g).to(DEV) * 0.1
    scale = (torch.rand(64, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(64, generator=g) * 0.3).to(DEV)
    x = torch.randn((B, 3, H, W), generator=g).to(DEV)
    x[0, 0, H // 2, W // 2] = float("inf")
    x[B - 1, 2, 1, 1] = float("nan")
    outs = []
    for fuse in (False, True):
        b = engine.ProgramBuilder(H, W, dtype)
        b.fuse_stem = fuse
        out = b.stem_pool("input", w, scale, shift)
        prog = b.p
        assert [op.kind for op in prog.ops] == (["stem7"] if fuse else ["to_nhwc4", "conv", "maxpool"])
        bufs = dict(prog._alloc(B, x.device))
        bufs["input"] = x
        for op in prog.ops:
            prog._launch(_lib.lib(), op, bufs, B, _lib.current_stream())
        torch.cuda.synchronize()
        outs.append(bufs[out].clone())
    hp, wp = ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1
    assert outs[0].numel() == B * hp * wp * 64
    assert torch.equal(outs[0].view(torch.int16 if dtype == "bf16" else torch.int32), outs[1].view(torch.int16 if dtype == "bf16" else torch.int32))
    # (the ReLU `v > 0 ? v : 0` of every path turns a NaN accumulator into 0; +inf survives it)
    assert float(torch.nan_to_num(outs[1].float(), nan=0.0, posinf=0.0).abs().max()) > 0 and bool(torch.isinf(outs[1].float()).any())


def test_full_batch_128_is_consistent_with_golden(golden):
    """BASELINE configs[1] size (bs=128): images repeat the two golden inputs, so every output must equal the
    golden pair's - bitwise among replicas (deterministic kernels), 1e-4 rel against the reference."""
    g = golden("g1_dconv_fwd.npz")
    m = _load(pose_resnet_dconv, "dconv", int(g["seed"]))
    x2 = synth.input_images(2, int(g["seed"]))
    x = _cuda(np.concatenate([x2] * 64, 0))
    with torch.no_grad():
        hm = m(x)
        hm2 = m(_cuda(x2))
    assert hm.shape == (128, 17, 64, 48)
    assert torch.equal(hm[0::2], hm[0:1].expand(64, -1, -1, -1)) and torch.equal(hm[1::2], hm[1:2].expand(64, -1, -1, -1))
    assert np.abs(hm[:2].cpu().numpy() - g["heat_maps"]).max() / np.abs(g["heat_maps"]).max() <= 1e-4
    assert np.abs(hm2.cpu().numpy() - g["heat_maps"]).max() / np.abs(g["heat_maps"]).max() <= 1e-4
    # reload different weights -> program is re-packed
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    with torch.no_grad():
        hm3 = m(_cuda(x2))
    assert not torch.allclose(hm3, hm2)


def test_program_run_writes_into_a_caller_owned_result_and_refuses_a_wrong_one(golden):
    g = golden("g1_dconv_fwd.npz")
    m = _load(pose_resnet_dconv, "dconv", int(g["seed"]))
    x = _cuda(synth.input_images(2, int(g["seed"])))
    prog = m.hip_program(x)
    out = torch.full((2, 17, 64, 48), float("nan"), device=DEV)
    ret = prog.run(x, out=out)
    assert ret.data_ptr() == out.data_ptr() and torch.equal(out, prog.run(x))
    with pytest.raises(ValueError):
        prog.run(x, out=torch.empty((2, 17, 64, 47), device=DEV))
    with pytest.raises(ValueError):
        prog.run(x, out=torch.empty((2, 17, 64, 48), device=DEV, dtype=torch.float64))


def test_empty_batches_give_empty_results_without_a_launch(golden):
    """An image without detections reaches every stage of the detector-driven path as a batch of ZERO (crop_boxes -> forward_crops ->
    decode -> rescoring + OKS-NMS): the reference's torch / numpy ops return empty tensors there, and so do the mirrors - no launch,
    no error."""
    from simple_pose_amd.datasets.coco import normalize_crops
    from simple_pose_amd.datasets.naive_data import crop_boxes, filter_poses, oks_nms
    from simple_pose_amd.metrics.pose_metrics import kps_to_dict_
    g = golden("g1_dconv_fwd.npz")
    m = _load(pose_resnet_dconv, "dconv", int(g["seed"]))
    img = torch.zeros((480, 640, 3), dtype=torch.uint8, device=DEV)
    crops, tinv, centers, scales, areas = crop_boxes(img, np.zeros((0, 4), np.float32))
    assert crops.shape == (0, 256, 192, 3) and tinv.shape == (0, 2, 3) and centers.shape == (0, 2) and areas.shape == (0,)
    assert normalize_crops(crops).shape == (0, 3, 256, 192)
    with torch.no_grad():
        hm = m.forward_crops(crops)
        hm2 = m(torch.empty((0, 3, 256, 192), device=DEV))
    assert hm.shape == (0, 17, 64, 48) and hm2.shape == (0, 17, 64, 48)
    for dec in (GaussTaylorKeyPointDecoder(), BasicKeyPointDecoder()):
        kps, mv = dec(hm, tinv)
        assert kps.shape == (0, 17, 2) and mv.shape == (0, 17, 1)
    co, mv = BasicKeyPointDecoder.heat_map_to_axis(hm)
    assert co.shape == (0, 17, 2) and mv.shape == (0, 17, 1)
    t, w = RefineSimpleTransform.get_heat_map(torch.empty((0, 17, 3), device=DEV), 2.0, (48, 64))
    assert t.shape == (0, 17, 64, 48) and w.shape == (0, 17)
    out = []
    kps_to_dict_(kps, mv, [], out)
    assert out == []
    assert filter_poses(torch.empty((0, 17, 3), device=DEV), np.zeros(0), np.zeros(0, np.float32), []) == []
    assert oks_nms(np.zeros((0, 17, 3)), np.zeros(0), np.zeros(0), 0.9) == []
    torch.cuda.synchronize()


def test_masked_mse_vs_oracle():
    B, J, H, W = 6, 17, 64, 48
    pred = synth.tensor_normal(8, "mse/p", (B, J, H, W))
    tgt = synth.tensor_uniform(8, "mse/t", (B, J, H, W))
    mask = (synth.tensor_uniform(8, "mse/m", (B, J)) < 0.7).astype(np.float32)
    loss_ref, grad_ref = pose_oracle.masked_mse(pred, tgt, mask, want_grad=True)
    loss = torch.zeros(1, device=DEV)
    grad = torch.empty((B, J, H, W), device=DEV)
    ws = torch.empty(4096, dtype=torch.uint8, device=DEV)
    dp, dt, dm = _cuda(pred), _cuda(tgt), _cuda(mask)   # keep alive: the library does not own its operands
    _lib.check(_lib.lib().sp_masked_mse(_lib.ptr(dp), _lib.ptr(dt), _lib.ptr(dm), B, J, H * W,
                                        _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), _lib.current_stream()))
    assert abs(loss.item() - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
    assert np.abs(grad.cpu().numpy() - grad_ref).max() <= 1e-7 * np.abs(grad_ref).max() + 1e-12


def test_hrnet_w32_forward_vs_reference_golden(golden):
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    g = golden("g3_hrnet_w32_fwd.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        hm = m(_cuda(synth.input_images(1, int(g["seed"]))))
    ref = g["heat_maps"]
    assert hm.shape == (1, 17, 64, 48)
    rel = np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel <= 1e-4, rel
    # batch of 16 (autotuned tiles) reproduces the single image bitwise
    x16 = _cuda(np.repeat(synth.input_images(1, int(g["seed"])), 16, 0))
    with torch.no_grad():
        hm16 = m(x16)
    assert torch.equal(hm16[3], hm[0]) and torch.equal(hm16[15], hm[0])


def test_se_variant_forward_vs_reference_golden(golden):
    g = golden("g1s_dconv_se_fwd.npz")
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv", se=True), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        hm = m(_cuda(synth.input_images(1, int(g["seed"]))))
        hm32 = m(_cuda(np.repeat(synth.input_images(1, int(g["seed"])), 32, 0)))
    ref = g["heat_maps"]
    assert np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max() <= 1e-4
    assert torch.equal(hm32[17], hm[0])


def test_se_variant_in_bf16_vs_fp32_reference_golden(golden, measured):
    """reduction=True with bf16 operands (sp_global_avg_pool_nhwc_bf16, the two FCs as bf16 1x1 convs, sp_se_gate_add_relu_nhwc_bf16)
    against the fp32 reference heat maps: the bf16 bar of the plain DConv net (CPU autocast itself: 1.07e-2)."""
    g = golden("g1s_dconv_se_fwd.npz")
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv", se=True), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = "bf16"
    with torch.no_grad():
        hm = m(_cuda(synth.input_images(1, int(g["seed"]))))
        hm32 = m(_cuda(np.repeat(synth.input_images(1, int(g["seed"])), 32, 0)))
    ref = g["heat_maps"]
    rel = np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max()
    measured("bf16_se_heat_map_rel_err", rel, 2e-2)
    assert 1e-5 < rel <= 2e-2
    assert torch.equal(hm32[17], hm[0])


# ---------------------------------------------------------------------------------------------- bf16 operand path
def _bf16_round(t):
    return t.to(torch.bfloat16).float()


BF16_CASES = [("1x1_64_256", 2, 64, 16, 12, 256, 1, 1, 0), ("3x3_s1_64", 2, 64, 16, 12, 64, 3, 1, 1), ("3x3_s2_128", 2, 128, 16, 12, 128, 3, 2, 1),
              ("1x1_s2_256_512", 2, 256, 16, 12, 512, 1, 2, 0), ("3x3_512_ragged_M", 1, 512, 8, 6, 512, 3, 1, 1),
              ("3x3_32_32_hrnet", 2, 32, 16, 12, 32, 3, 1, 1), ("1x1_2048_512", 2, 2048, 8, 6, 512, 1, 1, 0), ("3x3_bigM_128", 8, 128, 32, 24, 128, 3, 1, 1)]


@pytest.mark.parametrize("case", BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_bf16_conv_vs_torch_on_rounded_operands(case):
    """SP_CONV_BF16: bf16 operands, fp32 accumulation.  Reference = fp64 conv of the SAME bf16-rounded operands, so the
    only differences are fp32 accumulation order and the final bf16 rounding of the output (2^-9 relative)."""
    name, B, Cin, H, W, Cout, k, s, p = case
    w = _bf16_round(torch.from_numpy(synth.tensor_normal(1, name + "/w", (Cout, Cin, k, k), std=(2.0 / (Cin * k * k)) ** 0.5)))
    x = _bf16_round(torch.from_numpy(synth.tensor_normal(1, name + "/x", (B, Cin, H, W))))
    scale = torch.from_numpy(synth.tensor_uniform(1, name + "/s", (Cout,), 0.5, 1.5))
    shift = torch.from_numpy(synth.tensor_normal(1, name + "/b", (Cout,), std=0.3))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=s, padding=p)
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    res = _bf16_round(torch.from_numpy(synth.tensor_normal(1, name + "/r", tuple(ref.shape))))
    ref = torch.relu(ref + res.double())
    bld = engine.ProgramBuilder(H, W, dtype="bf16")
    bld.p.shapes["input"] = (H, W, Cin)
    bld.p.shapes["res"] = tuple(ref.shape[2:]) + (Cout,)
    out = bld.conv("input", w.to(DEV), stride=s, pad=p, scale=scale.to(DEV), shift=shift.to(DEV), relu=True, res="res")
    prog = bld.p
    prog.out_name = out
    prog.out_shape = prog.shapes[out]
    pool = prog._alloc(B, torch.device(DEV))
    pool["res"] = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)
    # run by hand: the program's generic run() allocates an fp32 output for out_name; here the output is a bf16 NHWC buffer
    y = torch.empty((B,) + tuple(prog.out_shape), dtype=torch.bfloat16, device=DEV)
    op = prog.ops[-1]
    op.desc.batch = B
    _lib.check(_lib.lib().sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), _lib.ptr(op.scale), _lib.ptr(op.shift), _lib.ptr(pool["res"]),
                                        _lib.ptr(y), _lib.current_stream()))
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 6e-3, err            # output rounding to bf16: 2^-8 of the value


# pinned from measured_parity.json, round 2: 1.28e-2 / 0.92e-2 / 1.45e-2 (deterministic: every tile and kernel gives the same bits);
# torch's CPU autocast-bf16 path against its own fp32: 1.07e-2
BF16_FWD_BAR = {"dconv": 1.5e-2, "duc": 1.5e-2, "hrnet_w32": 2e-2}


@pytest.mark.parametrize("arch", ["dconv", "duc", "hrnet_w32"])
def test_bf16_forward_vs_fp32_reference_golden(golden, measured, arch):
    """BASELINE configs 3 and 5 (bf16 compute): heat maps against the fp32 reference within the bf16 tolerance the CPU
    autocast path itself shows (1.07e-2 relative), and against our own fp32 path."""
    import os
    if arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        g = golden("g3_hrnet_w32_fwd.npz")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), int(g["seed"]))
    else:
        g = golden({"dconv": "g1_dconv_fwd.npz", "duc": "g2_duc_fwd.npz"}[arch])
        m = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[arch].resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(arch), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    B = int(g["batch"])
    x = _cuda(synth.input_images(B, int(g["seed"])))
    with torch.no_grad():
        hm32 = m(x)
        m.compute_dtype = "bf16"
        hm16 = m(x)
    assert hm16.dtype == torch.float32 and hm16.shape == hm32.shape
    ref = g["heat_maps"]
    rel = np.abs(hm16.cpu().numpy() - ref).max() / np.abs(ref).max()
    measured("bf16_heat_map_rel_err", rel, BF16_FWD_BAR[arch])
    assert 1e-5 < rel <= BF16_FWD_BAR[arch], rel
    assert np.abs(hm32.cpu().numpy() - ref).max() / np.abs(ref).max() <= 1e-4     # switching back and forth re-packs


@pytest.mark.parametrize("arch", ["dconv", "duc", "hrnet_w32"])
def test_full_batch_128_bf16_is_consistent_with_golden(golden, measured, arch):
    """BASELINE configs 3 and 5 at their full size (bs=128, bf16): the batch repeats the golden inputs, so every output must equal
    its replica bit for bit (deterministic kernels, tile- and kernel-independent reduction order: the autotuned bs=128 table
    mixes ring / implicit-GEMM / direct kernels), equal the small-batch run of the same images bit for bit, and stay within
    the bf16 bar of the fp32 reference."""
    import os
    if arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        g = golden("g3_hrnet_w32_fwd.npz")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), int(g["seed"]))
    else:
        g = golden({"dconv": "g1_dconv_fwd.npz", "duc": "g2_duc_fwd.npz"}[arch])
        m = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[arch].resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(arch), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = "bf16"
    n = int(g["batch"])
    xs = synth.input_images(n, int(g["seed"]))
    x = _cuda(np.concatenate([xs] * (128 // n), 0))
    with torch.no_grad():
        hm = m(x)                      # autotunes at bs=128
        hms = m(_cuda(xs))
    assert hm.shape == (128, 17, 64, 48)
    for i in range(n):
        assert torch.equal(hm[i::n], hm[i:i + 1].expand(128 // n, -1, -1, -1)), (arch, i)
    assert torch.equal(hm[:n], hms), arch
    rel = np.abs(hm[:n].cpu().numpy() - g["heat_maps"]).max() / np.abs(g["heat_maps"]).max()
    measured("bf16_bs128_heat_map_rel_err", rel, BF16_FWD_BAR[arch])
    assert rel <= BF16_FWD_BAR[arch], rel


def test_heat_map_acc_and_collate_normalisation_vs_reference_golden(golden):
    from simple_pose_amd.datasets.coco import normalize_crops
    from simple_pose_amd.metrics.pose_metrics import HeatMapAcc
    g = golden("g7_next.npz")
    acc = HeatMapAcc()
    for tag in ("a", "b"):
        tgt, _ = RefineSimpleTransform.get_heat_map(_cuda(g[f"acc/{tag}/joints"]), 2.0, (48, 64))
        val = acc(_cuda(g[f"acc/{tag}/pred"]), tgt)
        assert val.is_cuda and val.dim() == 0
        assert abs(val.item() - float(g[f"acc/{tag}/value"])) < 1e-6
    # the solver's masked form (ddp...:130-131): a mask argument == multiplying both maps by the mask
    pred, joints = _cuda(g["acc/a/pred"]), g["acc/a/joints"]
    tgt, w = RefineSimpleTransform.get_heat_map(_cuda(joints), 2.0, (48, 64))
    mask = w.clone()
    mask[0, ::3] = 0.0
    m4 = mask[..., None, None]
    assert acc(pred, tgt, mask).item() == acc(pred * m4, tgt * m4).item()
    assert acc(pred, tgt, mask).item() != acc(pred, tgt).item() or float(mask.min()) > 0
    x = normalize_crops(_cuda(g["collate/img_u8"]))
    assert np.array_equal(x.cpu().numpy(), g["collate/input"])          # bit exact: x/255 - mean in fp32


def test_oks_nms_and_result_scores_vs_reference_golden(golden):
    """eval.py:153-197 + naive_data.py:120-173 + kps_to_dict_ on the GPU vs the real reference's outputs."""
    from simple_pose_amd.datasets.naive_data import filter_poses, oks_nms
    from simple_pose_amd.metrics.pose_metrics import kps_to_dict_
    g = golden("g8_nms.npz")
    for tag in "abc":
        vis, thr = g[f"filter/{tag}/params"]
        res = filter_poses(_cuda(g["kps"]), g["box_score"], g["area"], g["img_id"].tolist(), in_vis_thre=float(vis), oks_thre=float(thr))
        assert [r["image_id"] for r in res] == g[f"filter/{tag}/image_id"].tolist()
        np.testing.assert_array_equal(np.array([r["keypoints"] for r in res]), g[f"filter/{tag}/keypoints"])   # same persons, same order
        np.testing.assert_allclose(np.array([r["score"] for r in res]), g[f"filter/{tag}/score"], rtol=1e-15)
    k64, a64 = g["kps"][23:].astype(np.float64), g["area"][23:].astype(np.float64)
    assert oks_nms(k64, g["direct/scores"], a64, 0.6, None, 0.3) == g["direct/keep_vis"].tolist()
    assert oks_nms(k64, g["direct/scores"], a64, 0.8, g["direct/sigmas"], None) == g["direct/keep_sig"].tolist()
    lst = []
    kps_to_dict_(_cuda(g["kps"][:8, :, :2].copy()), _cuda(g["kps"][:8, :, 2:].copy()), g["dict/image_id"].tolist(), lst)
    np.testing.assert_allclose([d["score"] for d in lst], g["dict/score"], rtol=2e-7)
    np.testing.assert_array_equal(np.array([d["keypoints"] for d in lst]), g["dict/keypoints"])
    assert [d["image_id"] for d in lst] == g["dict/image_id"].tolist()


def test_oks_nms_large_batch_vs_oracle():
    """256 images x up to 200 persons in one launch against the C oracle image by image; includes an image with equal scores."""
    from simple_pose_amd.datasets.naive_data import oks_nms_batch
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 200, 256)
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    Pn = int(seg[-1])
    base = rng.random((Pn // 4 + 1, 17, 3)) * np.array([400.0, 600.0, 1.0])
    kps = base[rng.integers(0, len(base), Pn)] + rng.normal(size=(Pn, 17, 3)) * np.array([3.0, 3.0, 0.03])
    scores, areas = rng.random(Pn), rng.random(Pn) * 30000 + 800
    scores[seg[3]:seg[4]] = 0.5
    keep, cnt = oks_nms_batch(torch.from_numpy(kps).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(areas).cuda(),
                              torch.from_numpy(seg).cuda(), int(sizes.max()), 0.7)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    n_kept = 0
    for gi in range(256):
        lo, hi = seg[gi], seg[gi + 1]
        ref = pose_oracle.oks_nms(kps[lo:hi], scores[lo:hi], areas[lo:hi], 0.7)
        assert (keep[lo:lo + cnt[gi]] - lo).tolist() == ref, gi
        assert (keep[lo + cnt[gi]:hi] == -1).all()
        n_kept += len(ref)
    assert 0 < n_kept < Pn


def test_captured_hip_graph_replays_forward_and_decode_bitwise(golden):
    """Program.capture: forward + GaussTaylor decode of one batch shape as ONE hipGraph; replays must equal the stream launches
    bit for bit, for inputs written into the graph's static buffers after capture."""
    g = golden("g1_dconv_fwd.npz")
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), int(g["seed"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    dec = GaussTaylorKeyPointDecoder()
    B = int(g["heat_maps"].shape[0])
    x0 = _cuda(synth.input_images(B, int(g["seed"])))
    x1 = _cuda(synth.input_images(B, 77))
    tinv = _cuda(synth.trans_inv_batch(B))
    prog = net.hip_program(x0)
    graphed = prog.capture(x1, dec, tinv)                      # captured on OTHER data than what is replayed first
    for x in (x0, x1, x0):
        hm_e = prog.run(x)
        kps_e, mv_e = dec(hm_e, tinv)
        hm_g, kps_g, mv_g = graphed(x)
        torch.cuda.synchronize()
        assert torch.equal(hm_g, hm_e) and torch.equal(kps_g, kps_e) and torch.equal(mv_g, mv_e)
    rel = np.abs(graphed(x0)[0].cpu().numpy() - g["heat_maps"]).max() / np.abs(g["heat_maps"]).max()
    assert rel <= 1e-4, rel


def test_pipelined_decode_equals_the_in_line_decode_bitwise():
    """engine.PipelinedForward: the decode of batch i on its own stream under the forward of batch i + 1, two alternating heat-map buffers -
    key points and scores of five consecutive (different) batches equal the in-line forward + decode bit for bit."""
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 4)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    net.autotune = False
    dec = GaussTaylorKeyPointDecoder()
    xs = [_cuda(synth.input_images(6, 30 + i, h=128, w=96)) for i in range(5)]
    tinv = _cuda(synth.trans_inv_batch(6))
    prog = net.hip_program(xs[0])
    ref = []
    for x in xs:
        k, m = dec(prog.run(x), tinv)
        ref.append((k.clone(), m.clone()))
    piped = engine.PipelinedForward(prog, dec)
    got = [piped(x, tinv) for x in xs]
    piped.sync()
    torch.cuda.synchronize()
    for (k, m), (rk, rm) in zip(got, ref):
        assert torch.equal(k, rk) and torch.equal(m, rm)


def test_interleaved_forward_equals_run_bitwise():
    """engine.InterleavedForward: consecutive batches on independent streams with their own activation pools, lane streams and events (HRNet:
    the forward of batch i + 1 under the low-occupancy tail of batch i) - key points and scores of seven different batches equal
    Program.run + decoder bit for bit, for depth 2 and 3."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    net.compute_dtype = "bf16"
    dec = GaussTaylorKeyPointDecoder()
    xs = [_cuda(synth.input_images(4, 50 + i, h=128, w=96)) for i in range(7)]
    tinv = _cuda(synth.trans_inv_batch(4))
    prog = net.hip_program(xs[0])
    ref = []
    for x in xs:
        k, m = dec(prog.run(x), tinv)
        ref.append((k.clone(), m.clone()))
    for depth in (2, 3):
        inter = engine.InterleavedForward(prog, dec, depth=depth)
        got = [inter(x, tinv) for x in xs]
        inter.sync()
        torch.cuda.synchronize()
        for (k, m), (rk, rm) in zip(got, ref):
            assert torch.equal(k, rk) and torch.equal(m, rm)
        n_slots = sum(len(key) == 3 for key in prog._pools)
        assert n_slots == depth
        inter.close()
        assert not any(len(key) == 3 for key in prog._pools)


@pytest.mark.parametrize("mode", ["dconv_f32_bs128_depth2", "hrnet_bf16_single_stream_depth3"])
def test_interleaved_forward_in_the_modes_bench_defaults_to(mode):
    """bench.py's default measurement modes (round-3 verdict, weak 3): the headline = ResNet50-DConv fp32, 256x192, bs=128, two batches in
    flight; config 5 = HRNet-W32 bf16 with every forward on ONE stream (`multi_stream = False`) and three batches in flight.  Key points
    and scores of five different batches equal Program.run + decoder bit for bit (reference path: nets.*.forward +
    metrics/pose_metrics.py:55-107, one batch at a time)."""
    import os
    dec = GaussTaylorKeyPointDecoder()
    if mode.startswith("dconv"):
        net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
        sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 4)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net = net.cuda().eval()
        B, depth, n = 128, 2, 5
    else:
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=3)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        net = net.cuda().eval()
        net.compute_dtype = "bf16"
        B, depth, n = 32, 3, 7
    net.autotune = False
    base = [synth.input_images(8, 70 + i) for i in range(n)]           # n different batches: 8 distinct images each, rolled to B
    xs = [_cuda(np.roll(np.concatenate([b] * (B // 8), 0), i, axis=0)) for i, b in enumerate(base)]
    tinv = _cuda(synth.trans_inv_batch(B))
    prog = net.hip_program(xs[0])
    if not mode.startswith("dconv"):
        prog.multi_stream = False
    ref = []
    for x in xs:
        k, m = dec(prog.run(x), tinv)
        ref.append((k.clone(), m.clone()))
    inter = engine.InterleavedForward(prog, dec, depth=depth)
    got = [inter(x, tinv) for x in xs]
    inter.sync()
    torch.cuda.synchronize()
    for (k, m), (rk, rm) in zip(got, ref):
        assert torch.equal(k, rk) and torch.equal(m, rm)
    assert not torch.equal(ref[0][0], ref[1][0])                       # (the batches really differ)
    inter.close()


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_multi_term_fuse_kernel_vs_torch_and_the_chain(bf16):
    """sp_upsample_add_n_nhwc (HighResolutionModule.forward's `y = y + fuse_layers[i][j](x[j])` for j >= i, pose_hrnet.py:250-257, as one
    launch): fp32 = the chained sp_upsample_add_nhwc launches bit for bit; bf16 = the fp32 sum of the same operands rounded once."""
    import ctypes
    lib, st = _lib.lib(), _lib.current_stream()
    B, H, W, C = 3, 16, 24, 32
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator().manual_seed(5)
    base = torch.randn(B, H, W, C, generator=g).to(dt).to(DEV)
    for factors, relu in (((1,), 1), ((2,), 0), ((2, 4, 8), 1), ((1, 2, 4), 1), ((1, 2), 0)):
        terms = [torch.randn(B, H // f, W // f, C, generator=g).to(dt).to(DEV) for f in factors]
        y = torch.empty_like(base)
        xs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
        fs = (ctypes.c_int32 * len(terms))(*factors)
        _lib.check(lib.sp_upsample_add_n_nhwc(_lib.ptr(base), int(bf16), len(terms), xs, fs, _lib.ptr(y), B, H, W, C, relu, st), "fuse")
        ref = base.float()
        for t, f in zip(terms, factors):
            ref = ref + t.float().repeat_interleave(f, 1).repeat_interleave(f, 2)
        if relu:
            ref = ref.clamp(min=0)
        torch.cuda.synchronize()
        assert torch.equal(y, ref.to(dt))
        if not bf16:                                   # the chain it replaces: one launch per term
            cur = base
            for k, (t, f) in enumerate(zip(terms, factors)):
                nxt = torch.empty_like(base)
                _lib.check(lib.sp_upsample_add_nhwc(_lib.ptr(t), _lib.ptr(cur), _lib.ptr(nxt), B, H // f, W // f, C, f, int(relu and k == len(terms) - 1), st), "chain")
                cur = nxt
            torch.cuda.synchronize()
            assert torch.equal(cur, y)
    bad = (ctypes.c_int32 * 1)(3)                      # a factor that does not divide the output is refused, nothing is launched
    xs = (ctypes.c_void_p * 1)(base.data_ptr())
    assert lib.sp_upsample_add_n_nhwc(_lib.ptr(base), int(bf16), 1, xs, bad, _lib.ptr(base), B, H, W, C, 0, st) != 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hrnet_with_one_launch_per_fuse_output_vs_the_chained_program(dtype, measured):
    """HRNet-W32 with `fuse_terms` (23 fuse launches per forward instead of 43): fp32 heat maps equal the chained program's bit for bit;
    in bf16 the fused sums are rounded once instead of after every term - no further from the fp32 program than the chained one."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    net.compute_dtype = dtype
    x = _cuda(synth.input_images(3, 60))
    with torch.no_grad():
        net.fuse_terms = True
        prog = net.hip_program(x)
        n_fused = sum(op.kind == "upsample_add_n" for op in prog.ops)
        assert n_fused == 23 and not any(op.kind == "upsample_add" for op in prog.ops)
        a = net(x).clone()
        net.fuse_terms = False
        prog = net.hip_program(x)
        assert sum(op.kind == "upsample_add" for op in prog.ops) == 43
        b = net(x).clone()
    if dtype == "fp32":
        assert torch.equal(a, b)
    else:
        # two bf16 programs differ from each other by up to the sum of their own rounding noise (measured 2.7e-2 of the maximum); the
        # meaningful comparison is each against the fp32 program: rounding the fused sums once must not be worse than rounding per term
        with torch.no_grad():
            net.compute_dtype = "fp32"
            net.fuse_terms = True
            ref = net(x).clone()
        err_f = float((a - ref).abs().max() / ref.abs().max())
        err_c = float((b - ref).abs().max() / ref.abs().max())
        measured("bf16_fused_rel_err_vs_fp32", err_f, 2.5e-2)          # measured 2.01e-2 (chained: 2.21e-2) on these weights / images
        measured("bf16_chained_rel_err_vs_fp32", err_c)
        assert 1e-5 < err_f <= 2.5e-2 and err_f <= 1.05 * err_c, (err_f, err_c)


@pytest.mark.parametrize("B,H,W", [(2, 64, 48), (3, 34, 18), (1, 8, 6), (5, 32, 16)])
def test_hrnet_transition1_kernel_vs_float64(B, H, W, measured):
    """sp_hrnet_transition1 (transition1.0 = conv3x3 256 -> 32 stride 1 and transition1.1 = conv3x3 256 -> 64 stride 2, both + BN + ReLU, of
    the SAME input: nets/pose_hrnet.py:327-366, :431-437) in one launch against float64 on the same bf16-rounded operands - the bar of
    every other bf16 conv (output rounding) - on the network's 64x48 map, ragged tiles (34x18: one row / two columns past a tile) and
    maps smaller than one tile."""
    import ctypes
    lib, st = _lib.lib(), _lib.current_stream()
    assert lib.sp_hrnet_transition1_ok(256, H, W) == 1 and lib.sp_hrnet_transition1_ok(128, H, W) == 0
    x = torch.from_numpy(synth.tensor_normal(31, "t1/x", (B, 256, H, W))).to(torch.bfloat16)
    outs, refs = [], []
    packed = []
    for tag, cout, stride in (("a", 32, 1), ("b", 64, 2)):
        w = torch.from_numpy(synth.tensor_normal(31, f"t1/w{tag}", (cout, 256, 3, 3), std=(2.0 / 2304) ** 0.5)).to(torch.bfloat16)
        g, b_ = (torch.from_numpy(synth.tensor_uniform(31, f"t1/{n}{tag}", (cout,), 0.5, 1.5)) for n in "gb")
        m = torch.from_numpy(synth.tensor_normal(31, f"t1/m{tag}", (cout,), std=0.3))
        v = torch.from_numpy(synth.tensor_uniform(31, f"t1/v{tag}", (cout,), 0.2, 2.0))
        ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=1)
        refs.append(torch.relu(torch.nn.functional.batch_norm(ref, m.double(), v.double(), g.double(), b_.double(), False, 0.0, 1e-5)))
        n_pad, k_pad = ctypes.c_int(0), ctypes.c_int(0)
        assert lib.sp_conv_packed_dims(cout, 2304, 1, ctypes.byref(n_pad), ctypes.byref(k_pad)) == 0 and (n_pad.value, k_pad.value) == (cout, 2304)
        pk = torch.empty((cout, 2304), dtype=torch.bfloat16, device=DEV)
        _lib.check(lib.sp_pack_conv_weights(_lib.ptr(w.float().to(DEV)), cout, 256, 3, 3, 256, 3, 0, -1, cout, 2304, _lib.ptr(pk), 1, st))
        scale, shift = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
        dg, db, dm, dv = (t.to(DEV) for t in (g, b_, m, v))
        _lib.check(lib.sp_fold_bn(_lib.ptr(dg), _lib.ptr(db), _lib.ptr(dm), _lib.ptr(dv), cout, 1e-5, 0, _lib.ptr(scale), _lib.ptr(shift), st))
        packed.append((pk, scale, shift))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    ya = torch.full((B, H, W, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    yb = torch.full((B, H // 2, W // 2, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    (pa, sa, ha), (pb, sb, hb) = packed
    _lib.check(lib.sp_hrnet_transition1(_lib.ptr(xn), B, H, W, _lib.ptr(pa), 2304, _lib.ptr(sa), _lib.ptr(ha), _lib.ptr(pb), _lib.ptr(sb), _lib.ptr(hb),
                                        _lib.ptr(ya), _lib.ptr(yb), st), "transition1")
    torch.cuda.synchronize()
    for tag, y, ref in (("hi", ya, refs[0]), ("lo", yb, refs[1])):
        got = y.float().cpu().permute(0, 3, 1, 2).double()
        assert not torch.isnan(got).any(), tag                          # every output element written
        err = float((got - ref).abs().max() / ref.abs().max())
        measured(f"transition1_{tag}_rel_err", err, 6e-3)
        assert err <= 6e-3, (tag, err)
    assert lib.sp_hrnet_transition1(_lib.ptr(xn), B, H + 1, W, _lib.ptr(pa), 2304, _lib.ptr(sa), _lib.ptr(ha), _lib.ptr(pb), _lib.ptr(sb), _lib.ptr(hb),
                                    _lib.ptr(ya), _lib.ptr(yb), st) != 0          # odd sizes are refused


def test_hrnet_with_the_fused_transition_vs_the_two_conv_launches(measured):
    """HRNet-W32 bf16 with transition1 as one `htrans` op against the program with its two conv launches: the heat maps agree to bf16
    noise (another fp32 summation order in two layers), each within the bf16 bar of the fp32 program."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    x = _cuda(synth.input_images(3, 60))
    with torch.no_grad():
        net.compute_dtype = "fp32"
        ref = net(x).clone()
        net.compute_dtype = "bf16"
        net.fuse_transition = True
        prog = net.hip_program(x)
        assert sum(op.kind == "htrans" for op in prog.ops) == 1 and not any(op.name.startswith("transition1") and op.kind == "conv" for op in prog.ops)
        a = net(x).clone()
        net.fuse_transition = False
        assert not any(op.kind == "htrans" for op in net.hip_program(x).ops)
        b = net(x).clone()
    err_f = float((a - ref).abs().max() / ref.abs().max())
    err_c = float((b - ref).abs().max() / ref.abs().max())
    measured("bf16_fused_transition_rel_err_vs_fp32", err_f, 2.5e-2)
    measured("bf16_two_launch_rel_err_vs_fp32", err_c)
    assert 1e-5 < err_f <= 2.5e-2 and err_f <= 1.15 * err_c, (err_f, err_c)


def test_captured_graph_survives_the_eviction_of_its_activation_pool():
    """Program._alloc keeps MAX_POOLS activation pools; a captured hipGraph has the pointers of ITS pool baked in, so it must keep that
    pool alive: capture at one batch size, run more other batch sizes than pools are kept (which evicts the captured size from the
    program's table), scribble over freed memory, replay - still bit-identical to the stream launches."""
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 4)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    net.autotune = False
    x = _cuda(synth.input_images(4, 12, h=128, w=96))
    prog = net.hip_program(x)
    ref = prog.run(x).clone()
    graphed = prog.capture(x)
    assert (4, str(x.device)) in prog._pools
    for b in (2, 6, 3):                                        # MAX_POOLS = 2: the pool of batch 4 leaves the table
        prog.run(_cuda(synth.input_images(b, b, h=128, w=96)))
    assert (4, str(x.device)) not in prog._pools and len(prog._pools) <= prog.MAX_POOLS
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]       # whatever the allocator hands out now is poisoned
    torch.cuda.synchronize()
    out = graphed(x)
    out = out[0] if isinstance(out, tuple) else out
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    del junk


def test_gpu_person_crops_vs_reference_glue_golden(golden):
    """crop_boxes (one launch for all boxes of an image) == the crops / trans_inv the reference's BasicTransform
    produced through the restated OpenCV primitives, bit for bit; feeds normalize_crops -> model as eval.py's loader would."""
    from simple_pose_amd.datasets.coco import normalize_crops
    from simple_pose_amd.datasets.naive_data import crop_boxes
    g = golden("g9_crop.npz")
    crops, tinv, centers, scales, areas = crop_boxes(_cuda(g["img"]), g["boxes"])
    np.testing.assert_array_equal(crops.cpu().numpy(), g["crops"])
    np.testing.assert_array_equal(tinv.cpu().numpy(), g["trans_inv"].astype(np.float32))
    np.testing.assert_array_equal(centers, g["centers"]); np.testing.assert_array_equal(scales, g["scales"])
    np.testing.assert_array_equal(areas, g["areas"])
    x = normalize_crops(crops)
    assert x.shape == (5, 3, 256, 192) and x.dtype == torch.float32
    # a large, rotated, partly outside warp against the C oracle
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    th = 0.4
    M = np.array([[[1.3 * np.cos(th), -1.3 * np.sin(th), -80.5], [1.3 * np.sin(th), 1.3 * np.cos(th), -140.25]],
                  [[0.31, 0.0, 10.0], [0.0, 0.29, -3.0]], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], [[4.0, 0.0, -2000.0], [0.0, 4.0, 5.0]]])
    out = torch.empty((4, 256, 192, 3), dtype=torch.uint8, device="cuda")
    imd = torch.from_numpy(img).cuda()
    M = np.ascontiguousarray(M)
    _lib.check(_lib.lib().sp_warp_affine_u8c3(_lib.ptr(imd), 480, 640, M.ctypes.data, 4, _lib.ptr(out), 256, 192, _lib.current_stream()))
    for i in range(4):
        np.testing.assert_array_equal(out[i].cpu().numpy(), pose_oracle.warp_affine_u8c3(img, M[i], (192, 256)))
    np.testing.assert_array_equal(out[2].cpu().numpy(), img[:256, :192])                 # identity map = plain copy


def _sweep_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        k = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 1, 2]))
        cin = int(rng.choice([32, 64, 96, 128, 160, 256]))
        cout = int(rng.choice([17, 32, 48, 64, 100, 128, 192, 256, 320]))
        B, H, W = int(rng.integers(1, 6)), int(rng.integers(3, 23)), int(rng.integers(3, 19))
        p = int(rng.integers(0, 2)) if k == 3 else 0
        if (H + 2 * p - k) // s + 1 < 1 or (W + 2 * p - k) // s + 1 < 1:
            continue
        cases.append((B, cin, H, W, cout, k, s, p))
    return cases


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv_random_ragged_shapes_on_every_tile(dtype):
    """20 random layer shapes (odd spatial sizes, M and N that are no multiple of any tile, channels that are no multiple of 64,
    pad 0/1, stride 1/2) x every workgroup tile the launcher accepts for them: all tiles must agree bit for bit with each other
    (same K order) and with an fp64 convolution to fp32 / bf16-operand accuracy."""
    for ci, (B, Cin, H, W, Cout, k, s, p) in enumerate(_sweep_cases(20, 2024)):
        tag = f"sweep{ci}"
        if dtype == "bf16":
            Cout = (Cout + 7) // 8 * 8                       # bf16 NHWC stores are 16-byte: c_out % 8 == 0 is part of the contract
        w = torch.from_numpy(synth.tensor_normal(3, tag + "/w", (Cout, Cin, k, k), std=(2.0 / (Cin * k * k)) ** 0.5))
        x = torch.from_numpy(synth.tensor_normal(3, tag + "/x", (B, Cin, H, W)))
        shift = torch.from_numpy(synth.tensor_normal(3, tag + "/b", (Cout,), std=0.3))
        if dtype == "bf16":
            w, x = w.bfloat16().float(), x.bfloat16().float()          # exact operands for the reference
        ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), stride=s, padding=p) + shift.double().view(1, -1, 1, 1))
        b = engine.ProgramBuilder(H, W, dtype=dtype)
        b.p.shapes["input"] = (H, W, Cin)
        out = b.conv("input", w.to(DEV), stride=s, pad=p, shift=shift.to(DEV), relu=True, name="c")
        prog = b.p
        prog.out_name, prog.out_shape = out, prog.shapes[out]
        op = [o for o in prog.ops if o.kind == "conv"][0]
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(tdt)
        op.desc.batch = B
        results = []
        for tm, tn in _lib.CONV_TILES:
            if op.desc.n_pad % tn:
                continue
            op.desc.tile_m, op.desc.tile_n = tm, tn
            y = torch.full((B,) + tuple(prog.out_shape), float("nan"), dtype=tdt, device=DEV)      # every element must be written
            _lib.check(_lib.lib().sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), _lib.ptr(op.scale), _lib.ptr(op.shift), None,
                                                _lib.ptr(y), _lib.current_stream()))
            torch.cuda.synchronize()
            results.append(((tm, tn), y.float().cpu()))
        assert len(results) >= (2 if op.desc.n_pad >= 64 else 1), (Cout, op.desc.n_pad)
        for tile, y in results[1:]:
            assert torch.equal(y, results[0][1]), (ci, tile, (B, Cin, H, W, Cout, k, s, p))
        got = results[0][1].permute(0, 3, 1, 2).double()
        err = (got - ref).abs().max() / ref.abs().max()
        assert err < (2e-6 if dtype == "fp32" else 6e-3), (ci, err, (B, Cin, H, W, Cout, k, s, p))


def test_ring_kernel_matches_igemm_on_ragged_shapes():
    """conv_ring.hip (persistent 8-wave workgroups, LDS-DMA ring; sp_conv_desc.kernel = SP_CONV_KERNEL_RING) against the register-staged
    implicit GEMM on shapes that stress what is new in it: M that is no multiple of any tile and smaller than one tile (rows beyond M,
    fewer tiles than CUs), several output tiles per workgroup (the K-tile stream crossing tile boundaries), K of exactly the ring
    depth, stride 2, residual + ReLU, the fused PixelShuffle store and the 4 phases of the transposed conv.  Every ring tile the
    library accepts must give the igemm kernel's bits, and both the fp64 convolution of the same bf16 operands."""
    lib = _lib.lib()
    cases = [  # (B, Cin, H, W, Cout, k, stride, pad, residual, kind)
        (3, 64, 13, 11, 64, 3, 1, 1, False, "conv"),
        (2, 128, 9, 7, 256, 1, 1, 0, True, "conv"),          # K = 2 tiles < every ring depth: no ring tile may accept it
        (5, 192, 17, 9, 128, 3, 2, 1, False, "conv"),
        (2, 256, 31, 23, 512, 1, 1, 0, True, "conv"),        # K = 4 tiles: the ring wraps inside every output tile
        (40, 64, 16, 12, 256, 3, 1, 1, True, "conv"),        # 7,680 rows x 256 columns: several tiles per workgroup on small tiles
        (2, 128, 12, 10, 512, 3, 1, 1, False, "pshuf"),
        (3, 256, 7, 5, 256, 4, 2, 1, False, "deconv"),
        (70, 512, 8, 6, 256, 4, 2, 1, False, "deconv"),      # 4 phases x 3,360 rows
    ]
    ran = ran_lw = ran_lw4 = 0
    for ci, (B, Cin, H, W, Cout, k, s, p, with_res, kind) in enumerate(cases):
        tag = f"ring{ci}"
        wshape = (Cin, Cout, 4, 4) if kind == "deconv" else (Cout, Cin, k, k)
        w = torch.from_numpy(synth.tensor_normal(5, tag + "/w", wshape, std=(2.0 / (Cin * k * k)) ** 0.5)).bfloat16().float()
        x = torch.from_numpy(synth.tensor_normal(5, tag + "/x", (B, Cin, H, W))).bfloat16().float()
        scale = torch.from_numpy(synth.tensor_uniform(5, tag + "/s", (Cout,), 0.5, 1.5)).float()
        shift = torch.from_numpy(synth.tensor_normal(5, tag + "/b", (Cout,), std=0.3))
        b = engine.ProgramBuilder(H, W, dtype="bf16")
        b.p.shapes["input"] = (H, W, Cin)
        res_name = None
        if kind == "deconv":
            out = b.deconv_k4s2p1("input", w.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), relu=True, name="c")
            ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
        else:
            ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=s, padding=p)
            if with_res:
                b.p.shapes["res"] = (ref.shape[2], ref.shape[3], Cout)
                res_name = "res"
            perm = TorchPacker.row_perm(Cout, "cpu") if kind == "pshuf" else torch.arange(Cout)
            out = b.conv("input", w.to(DEV), stride=s, pad=p, scale=scale[perm].to(DEV), shift=shift[perm].to(DEV), relu=True, res=res_name,
                         pixel_shuffle=(kind == "pshuf"), name="c")
        ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        r = None
        if with_res:
            r = torch.from_numpy(synth.tensor_normal(5, tag + "/r", tuple(ref.shape))).bfloat16()
            ref = ref + r.double()
        ref = torch.relu(ref)
        if kind == "pshuf":
            ref = torch.nn.functional.pixel_shuffle(ref, 2)
        prog = b.p
        op = [o for o in prog.ops if o.kind == "conv"][0]
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
        rin = r.permute(0, 2, 3, 1).contiguous().to(DEV) if r is not None else None
        op.desc.batch = B
        results = []
        for cand in prog._candidates(lib, op):
            if cand[0] < 0:
                continue
            op.desc.tile_m, op.desc.tile_n, op.desc.kernel = cand
            y = torch.full((B,) + tuple(prog.shapes[out]), float("nan"), dtype=torch.bfloat16, device=DEV)
            _lib.check(lib.sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), _lib.ptr(op.scale), _lib.ptr(op.shift), _lib.ptr(rin), _lib.ptr(y),
                                         _lib.current_stream()), f"{tag} {cand}")
            torch.cuda.synchronize()
            results.append((cand, y.view(torch.int16).cpu(), y.float().cpu()))
        ring = [c for c, _, _ in results if c[2] == _lib.SP_CONV_KERNEL_RING]
        ran += len(ring)
        ran_lw += len([c for c, _, _ in results if c[2] == _lib.SP_CONV_KERNEL_RING_LW])      # the same ring fed by four loader waves (round 5)
        ran_lw4 += len([c for c, _, _ in results if c[2] == _lib.SP_CONV_KERNEL_RING_LW4])    # ... with four MFMA waves, one per SIMD (round 6)
        for cand, bits, _ in results[1:]:
            assert torch.equal(bits, results[0][1]), (tag, cand, int((bits != results[0][1]).sum()))
        got = results[0][2].permute(0, 3, 1, 2).double()
        assert not torch.isnan(got).any(), tag                  # every output element was written
        err = (got - ref).abs().max() / ref.abs().max()
        assert err < 6e-3, (tag, err)
    assert ran >= 20, ran                                      # the ring kernel really took part
    assert ran_lw >= 15, ran_lw                                # ... and so did its loader-wave variant
    assert ran_lw4 >= 20, ran_lw4                              # ... and the four-MFMA-wave tiles (96-row tiles included)
    print(f"ring-vs-igemm: {ran} ring launches + {ran_lw} loader-wave ring launches bit-identical to the implicit GEMM")


@pytest.mark.parametrize("Cin,Cout,B,H,W,with_res,relu", [(64, 256, 3, 16, 12, True, True), (64, 256, 2, 9, 7, False, False), (128, 512, 5, 8, 6, True, True),
                                                       (128, 256, 1, 5, 13, True, False), (64, 512, 40, 16, 12, False, True)])
def test_streaming_pointwise_kernel_equals_the_tiled_gemm_bitwise(Cin, Cout, B, H, W, with_res, relu):
    """kernel = SP_CONV_KERNEL_PW (csrc/conv_pw.hip: fp32 1x1, K = 64 / 128, c_out % 256 == 0 - the bottleneck's conv3 and the projection
    shortcut of layer1 / layer2, pose_resnet_dconv.py:99-103,124-131): one persistent workgroup per CU, weights in registers.  Same bits
    as every tile of the implicit GEMM, incl. row counts that are not multiples of its 64-row tile and more row tiles than workgroups;
    against float64 within the fp32 bar of the other conv tests."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(Cin + Cout + B)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    r = torch.randn((B, Cout, H, W), generator=g) if with_res else None
    b = engine.ProgramBuilder(H, W, "fp32")
    b.p.shapes["input"] = (H, W, Cin)
    if with_res:
        b.p.shapes["res"] = (H, W, Cout)
    out = b.conv("input", w.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), relu=relu, res="res" if with_res else None, name="c")
    op = b.p.ops[-1]
    op.desc.batch = B
    assert lib.sp_conv2d_pw_ok(op.desc) == 1
    cands = b.p._candidates(lib, op)
    assert (64, 256, _lib.SP_CONV_KERNEL_PW) in cands
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    rin = r.permute(0, 2, 3, 1).contiguous().to(DEV) if with_res else None
    results = []
    for cand in cands:
        op.desc.tile_m, op.desc.tile_n, op.desc.kernel = cand
        y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        _lib.check(lib.sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), _lib.ptr(op.scale), _lib.ptr(op.shift), _lib.ptr(rin), _lib.ptr(y),
                                     _lib.current_stream()), str(cand))
        torch.cuda.synchronize()
        results.append((cand, y.view(torch.int32).cpu(), y.cpu()))
    for cand, bits, _ in results[1:]:
        assert torch.equal(bits, results[0][1]), (cand, int((bits != results[0][1]).sum()))
    ref = torch.nn.functional.conv2d(x.double(), w.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    got = results[-1][2].permute(0, 3, 1, 2).double()
    assert not torch.isnan(got).any()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    # a descriptor the kernel does not take is refused, not mis-run
    op.desc.kernel, op.desc.stride = _lib.SP_CONV_KERNEL_PW, 2
    assert lib.sp_conv2d_pw_ok(op.desc) == 0


def test_hrnet_branches_on_separate_streams_match_single_stream(golden):
    """engine.Program lanes: the independent branches of every HRNet module run on their own HIP stream, ordered by events
    (RAW on activations, WAR/WAW on the planner's recycled storage).  The result must equal the one-stream schedule bit for bit,
    run after run (a missing dependency shows up as a difference); a captured hipGraph keeps to one stream and agrees too."""
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    x = _cuda(synth.input_images(6, 3))
    prog = net.hip_program(x)
    assert max(op.lane for op in prog.ops) == 3
    prog.multi_stream = False
    ref = prog.run(x).clone()
    prog.multi_stream = True
    for _ in range(8):
        assert torch.equal(prog.run(x), ref)
    waits, records, tails = prog._plan_sync(6, x.device)
    assert sum(len(w) for w in waits) > 50 and set(tails) == {0, 1, 2, 3}
    graphed = prog.capture(x)
    for _ in range(3):
        assert torch.equal(graphed(x), ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_size_layers_sampled_reference_and_exact_linearity(dtype):
    """BASELINE size (bs=128) on the three largest layer shapes of ResNet50-DConv - too big for a dense CPU reference, so:
    (1) 300 randomly sampled output elements per layer against float64 dot products computed on the host (catches addressing
    errors that only exist at full size: byte offsets here reach 4e8); (2) size-independent exact properties: y(2x) == 2*y(x)
    bit for bit (power-of-two scaling commutes with every rounding) and every workgroup tile gives the same bits."""
    B = 128
    rng = np.random.default_rng(11)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    shapes = [("layer1.conv2", 64, 64, 48, 64, 3, 1, 1, "conv"), ("layer2.conv3", 128, 32, 24, 512, 1, 1, 0, "conv"),
              ("deconv6", 256, 32, 24, 256, 4, 2, 1, "deconv")]
    for name, cin, H, W, cout, k, s, p, kind in shapes:
        x = torch.from_numpy(synth.tensor_normal(5, name + "/x", (B, H, W, cin))).to(tdt)          # NHWC
        if kind == "conv":
            w = torch.from_numpy(synth.tensor_normal(5, name + "/w", (cout, cin, k, k), std=(2.0 / (cin * k * k)) ** 0.5))
        else:
            w = torch.from_numpy(synth.tensor_normal(5, name + "/w", (cin, cout, 4, 4), std=(8.0 / (cin * 16)) ** 0.5))
        w = w.to(tdt).float()
        b = engine.ProgramBuilder(H, W, dtype=dtype)
        b.p.shapes["input"] = (H, W, cin)
        out = b.conv("input", w.to(DEV), stride=s, pad=p, name="c") if kind == "conv" else b.deconv_k4s2p1("input", w.to(DEV), name="c")
        prog = b.p
        op = [o for o in prog.ops if o.kind == "conv"][0]
        oh, ow, oc = prog.shapes[out]
        op.desc.batch = B
        xd = x.to(DEV)

        def run(inp):
            y = torch.full((B, oh, ow, oc), float("nan"), dtype=tdt, device=DEV)
            _lib.check(_lib.lib().sp_conv2d_fwd(op.desc, _lib.ptr(inp), _lib.ptr(op.w), None, None, None, _lib.ptr(y), _lib.current_stream()))
            torch.cuda.synchronize()
            return y

        y = run(xd)
        assert torch.isfinite(y.float()).all()
        assert torch.equal(run(xd * 2), y * 2)                                               # exact linearity
        tiles = [t for t in _lib.CONV_TILES if op.desc.n_pad % t[1] == 0]
        keep = (op.desc.tile_m, op.desc.tile_n)
        for tm, tn in tiles[:3]:
            op.desc.tile_m, op.desc.tile_n = tm, tn
            assert torch.equal(run(xd), y), (name, tm, tn)
        op.desc.tile_m, op.desc.tile_n = keep
        yh, xh, wh = y.float().cpu().numpy(), x.float().numpy().astype(np.float64), w.numpy().astype(np.float64)
        worst = 0.0
        for _ in range(300):
            bi, oy, ox, co = (int(rng.integers(0, n)) for n in (B, oh, ow, oc))
            acc = 0.0
            if kind == "conv":
                for ky in range(k):
                    for kx in range(k):
                        iy, ix = oy * s - p + ky, ox * s - p + kx
                        if 0 <= iy < H and 0 <= ix < W:
                            acc += float(xh[bi, iy, ix] @ wh[co, :, ky, kx])
            else:                                            # ConvTranspose2d(4,2,1): oy = 2*iy - 1 + ky
                for ky in range(4):
                    for kx in range(4):
                        ty, tx = oy + 1 - ky, ox + 1 - kx
                        if ty % 2 == 0 and tx % 2 == 0 and 0 <= ty // 2 < H and 0 <= tx // 2 < W:
                            acc += float(xh[bi, ty // 2, tx // 2] @ wh[:, co, ky, kx])
            worst = max(worst, abs(float(yh[bi, oy, ox, co]) - acc) / (abs(acc) + 1.0))
        assert worst < (2e-6 if dtype == "fp32" else 8e-3), (name, worst)


def test_hrnet_w48_at_384x288_vs_oracle():
    """The reference's other HRNet configuration (nets/hrnet_w48.yaml, 48/96/192/384 channels) at its usual 384x288 input: no golden
    fixture of it is committed, so the checker is the forward oracle (the torch-CPU restatement that is pinned on W32)."""
    import os
    import yaml
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "simple_pose_amd", "nets", "hrnet_w48.yaml")
    m = get_pose_net(path, pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), 9)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    x = synth.input_images(2, 9, h=384, w=288)
    with torch.no_grad():
        hm = m(_cuda(x))
    assert hm.shape == (2, 17, 96, 72)
    with open(path) as fh:
        cfg = yaml.safe_load(fh)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = nets_oracle.hrnet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(x), cfg).numpy()
    rel = np.abs(hm.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel <= 1e-4, rel


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_uint8_crops_straight_into_the_network(dtype, golden):
    """model.forward_crops(uint8 BGR crops) == model(normalize_crops(crops)) bit for bit: the collate normalisation and the NHWC
    layout change are one launch (sp_u8hwc_bgr_to_nhwc), for the crops crop_boxes makes."""
    from simple_pose_amd.datasets.coco import normalize_crops
    from simple_pose_amd.datasets.naive_data import crop_boxes
    g = golden("g9_crop.npz")
    crops, *_ = crop_boxes(_cuda(g["img"]), g["boxes"])
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 4)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    net.compute_dtype = dtype
    with torch.no_grad():
        a = net.forward_crops(crops)
        b = net(normalize_crops(crops))
    assert a.shape == (5, 17, 64, 48) and torch.equal(a, b)


def test_detector_driven_pipeline_composes_end_to_end(golden):
    """eval.py's inference path with every stage on the GPU and nothing but the final dict list crossing to the host: image + boxes
    -> crop_boxes -> forward_crops -> GaussTaylor decode (trans_inv from the crop geometry) -> filter_poses (rescoring + OKS-NMS).
    Stage by stage the pieces are pinned elsewhere; here the chain is checked against the oracle chain fed with the same crops."""
    from simple_pose_amd.datasets.naive_data import crop_boxes, filter_poses
    g = golden("g9_crop.npz")
    boxes = np.concatenate([g["boxes"], g["boxes"][:3] + np.float32(1.5)])            # 3 near-duplicate detections
    crops, tinv, centers, scales, areas = crop_boxes(_cuda(g["img"]), boxes)
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv"), 4)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    with torch.no_grad():
        hm = net.forward_crops(crops)
    kps, mv = GaussTaylorKeyPointDecoder()(hm, tinv)
    det_score = np.linspace(0.9, 0.5, len(boxes))
    res = filter_poses(torch.cat([kps, mv], dim=-1), det_score, areas, [7] * len(boxes), in_vis_thre=0.0, oks_thre=0.9)
    # oracle chain on the same crops
    x = pose_oracle.normalize_crops(crops.cpu().numpy())
    with torch.no_grad():
        ohm = nets_oracle.resnet_dconv_forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(x)).numpy()
    assert np.abs(hm.cpu().numpy() - ohm).max() / np.abs(ohm).max() <= 1e-4
    okps, omv = pose_oracle.decode_gauss_taylor(hm.cpu().numpy(), tinv.cpu().numpy())     # decoder on the SAME maps
    assert np.abs(kps.cpu().numpy() - okps).max() <= 1e-3 and np.array_equal(mv.cpu().numpy(), omv)
    k3 = np.concatenate([kps.cpu().numpy(), mv.cpu().numpy()], -1).astype(np.float64)
    osc = pose_oracle.pose_rescore(k3, det_score, 0.0)
    keep = pose_oracle.oks_nms(k3, osc, areas.astype(np.float64), 0.9)
    assert [r["keypoints"] for r in res] == [k3[i].reshape(-1).tolist() for i in keep]
    assert 1 <= len(res) <= len(boxes) and all(r["image_id"] == 7 for r in res)


@pytest.mark.parametrize("k,pad,H,W", [(7, 3, 64, 96), (3, 1, 34, 20), (5, 2, 32, 64)])
def test_bf16_stem_on_pixel_pairs_vs_fp64(k, pad, H, W):
    """The bf16 stem reads the NHWC4 image as x-pairs (sp_conv_desc.stride_x = 1 for a pixel stride of 2; odd and even paddings shift
    the pair grid differently).  Reference: fp64 stride-2 convolution of the same bf16-rounded image and weights."""
    B = 3
    x = torch.from_numpy(synth.input_images(B, seed=8, h=H, w=W)).bfloat16().float()
    w = torch.from_numpy(synth.tensor_normal(6, f"stem{k}/w", (32, 3, k, k), std=(2.0 / (3 * k * k)) ** 0.5)).bfloat16().float()
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=pad))
    b = engine.ProgramBuilder(H, W, dtype="bf16")
    x4 = b.to_nhwc4("input")
    out = b.conv(x4, w.to(DEV), stride=2, pad=pad, relu=True, name="conv1")
    op = b.p.ops[-1]
    assert op.desc.stride_x == 1 and op.desc.in_w == W // 2 and op.desc.c_in == 8 and op.desc.k_pad < k * 8 * 8
    oh, ow, oc = b.p.shapes[out]
    xin = torch.empty((B, H, W, 4), dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.lib().sp_nchw_to_nhwc4_bf16(_lib.ptr(x.to(DEV)), _lib.ptr(xin), B, 3, H, W, _lib.current_stream()))
    y = torch.full((B, oh, ow, oc), float("nan"), dtype=torch.bfloat16, device=DEV)
    op.desc.batch = B
    _lib.check(_lib.lib().sp_conv2d_fwd(op.desc, _lib.ptr(xin), _lib.ptr(op.w), None, None, None, _lib.ptr(y), _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(xin[..., :3].float().cpu(), x.permute(0, 2, 3, 1)) and (xin[..., 3] == 0).all()
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 6e-3, err


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("B,H,W,res_on,relu,affine", [(2, 20, 17, True, True, True), (3, 8, 16, False, False, False), (1, 7, 5, True, False, True),
                                                      (5, 33, 47, False, True, True), (128, 64, 48, True, True, True), (128, 32, 24, True, True, True),
                                                      (300, 16, 8, False, True, True), (128, 16, 12, True, True, True)])
def test_direct_3x3_kernels_are_bit_identical_to_the_implicit_gemm(B, H, W, res_on, relu, affine, C):
    """sp_conv3x3_direct (halo tile through LDS once as linear padded pixel rows; 32 / 64 channels: the whole filter resident in LDS,
    128 channels: streamed per tap; persistent workgroups) keeps the implicit GEMM's reduction order, so it must reproduce sp_conv2d_fwd
    bit for bit - ragged tiles, image borders, several tiles per workgroup, with / without scale-shift, residual, ReLU."""
    lib, P = _lib.lib(), _lib.ptr
    w = torch.from_numpy(synth.tensor_normal(1, f"d/w{C}", (C, C, 3, 3), std=(2.0 / (9 * C)) ** 0.5))
    scale = torch.from_numpy(synth.tensor_uniform(1, f"d/s{C}", (C,), 0.5, 1.5)).to(DEV) if affine else None
    shift = torch.from_numpy(synth.tensor_normal(1, f"d/b{C}", (C,), std=0.3)).to(DEV) if affine else None
    b = engine.ProgramBuilder(H, W, dtype="bf16")
    b.p.shapes["input"] = (H, W, C)
    b.p.shapes["res"] = (H, W, C)
    b.conv("input", w.to(DEV), pad=1, scale=scale, shift=shift, relu=relu, res="res" if res_on else None, name="c")
    op = b.p.ops[-1]
    assert op.direct and lib.sp_conv3x3_direct_ok(op.desc) == 1
    d = op.desc
    d.batch = B
    assert _lib.conv_kernel_name(d, res_on, 3) == f"conv3x3_c{C}_tile_kernel"
    x = torch.from_numpy(synth.tensor_normal(2, "d/x", (B, H, W, C))).to(DEV).bfloat16()
    r = torch.from_numpy(synth.tensor_normal(2, "d/r", (B, H, W, C))).to(DEV).bfloat16()
    y0 = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    y1 = y0.clone()
    st, rp = _lib.current_stream(), (P(r) if res_on else None)
    _lib.check(lib.sp_conv2d_fwd(d, P(x), P(op.w), P(scale), P(shift), rp, P(y0), st))
    _lib.check(lib.sp_conv3x3_direct(d, P(x), P(op.w), P(scale), P(shift), rp, P(y1), st))
    torch.cuda.synchronize()
    assert torch.isfinite(y1.float()).all() and torch.equal(y0, y1)
    # and what it refuses
    d.c_out = 96
    assert lib.sp_conv3x3_direct_ok(d) == 0 and lib.sp_conv3x3_direct(d, P(x), P(op.w), None, None, None, P(y1), st) != 0


@pytest.mark.parametrize("B,H,W,J,relu", [(128, 64, 48, 17, False), (3, 20, 17, 17, False), (2, 16, 12, 32, True), (5, 33, 47, 5, False), (1, 7, 5, 17, False)])
def test_head_3x3_kernel_is_bit_identical_to_the_implicit_gemm(B, H, W, J, relu):
    """The DUC head's last layer, nn.Conv2d(128, J, 3, padding=1) + bias with the fp32 NCHW heat-map store (nets/pose_resnet_duc.py:172-177), through
    conv3x3_c128_head_kernel (round 6: halo tile in LDS, the whole filter resident, 16-byte NCHW stores straight from the accumulator) against
    sp_conv2d_fwd bit for bit: the BASELINE shape, ragged tiles and widths that are no multiple of 4, every J <= 32."""
    lib, P = _lib.lib(), _lib.ptr
    w = torch.from_numpy(synth.tensor_normal(1, f"hd/w{J}", (J, 128, 3, 3), std=(2.0 / (9 * 128)) ** 0.5))
    bias = torch.from_numpy(synth.tensor_normal(1, f"hd/b{J}", (J,), std=0.3)).to(DEV)
    b = engine.ProgramBuilder(H, W, dtype="bf16")
    b.p.shapes["input"] = (H, W, 128)
    b.conv("input", w.to(DEV), pad=1, shift=bias, relu=relu, out_nchw=True, name="final_layer")
    op = b.p.ops[-1]
    d = op.desc
    d.batch = B
    assert op.direct and lib.sp_conv3x3_direct_ok(d) == 1 and _lib.conv_kernel_name(d, False, 3) == "conv3x3_c128_head_kernel"
    x = torch.from_numpy(synth.tensor_normal(2, "hd/x", (B, H, W, 128))).to(DEV).bfloat16()
    y0 = torch.full((B, J, H, W), float("nan"), dtype=torch.float32, device=DEV)
    y1 = y0.clone()
    st = _lib.current_stream()
    _lib.check(lib.sp_conv2d_fwd(d, P(x), P(op.w), None, P(bias), None, P(y0), st))
    _lib.check(lib.sp_conv3x3_direct(d, P(x), P(op.w), None, P(bias), None, P(y1), st))
    torch.cuda.synchronize()
    assert torch.isfinite(y1).all() and torch.equal(y0.view(torch.int32), y1.view(torch.int32)), int((y0 != y1).sum())
    ref = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2).double(), w.bfloat16().double(), bias.cpu().double(), padding=1)
    if relu:
        ref = torch.relu(ref)
    assert float((y1.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-5
    assert lib.sp_conv3x3_direct(d, P(x), P(op.w), None, P(bias), P(y0), P(y1), st) != 0        # no residual form


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("B,H,W", [(2, 20, 17), (3, 8, 16), (1, 7, 5), (5, 33, 47), (128, 64, 48), (37, 9, 40), (2, 96, 72), (3, 17, 49), (1, 8, 48),
                                   (300, 16, 48), (1, 1, 1), (2, 23, 100), (128, 32, 24), (64, 48, 36), (3, 5, 25), (7, 4, 24)])
def test_fused_basic_block_c32_is_bit_identical_to_its_two_convs(B, H, W, C):
    """sp_basic_block_c32 / sp_basic_block_c64 (HRNet BasicBlock of the 32- / 64-channel branch in one launch: conv1 on the halo'd strip, t kept in LDS
    as bf16, zero outside the image, residual = the block input) against the two conv launches it replaces - bit for bit on ragged strips and image
    borders - and against the fp64 block on the same bf16 operands."""
    if C == 64 and B * H * W > 128 * 64 * 48 // 2:
        pytest.skip("the 64-channel cases stop at the size of HRNet's 32 x 24 maps at bs=128 (x4)")
    lib, P = _lib.lib(), _lib.ptr
    w1, w2 = (torch.from_numpy(synth.tensor_normal(3, f"bb/w{i}", (C, C, 3, 3), std=(2.0 / (9 * C)) ** 0.5)).bfloat16().float() for i in (1, 2))
    s1, s2 = (torch.from_numpy(synth.tensor_uniform(3, f"bb/s{i}", (C,), 0.5, 1.5)) for i in (1, 2))
    h1, h2 = (torch.from_numpy(synth.tensor_normal(3, f"bb/h{i}", (C,), std=0.3)) for i in (1, 2))
    x = torch.from_numpy(synth.tensor_normal(3, "bb/x", (B, C, H, W))).bfloat16()
    outs = []
    for fuse in (True, False):
        b = engine.ProgramBuilder(H, W, dtype="bf16")
        b.fuse_blocks = b.fuse_blocks64 = fuse
        b.p.shapes["input"] = (H, W, C)
        y = b.basic_block_c32("input", w1.to(DEV), s1.to(DEV), h1.to(DEV), w2.to(DEV), s2.to(DEV), h2.to(DEV), name="blk")
        if not fuse:
            assert y is None
            t = b.conv("input", w1.to(DEV), pad=1, scale=s1.to(DEV), shift=h1.to(DEV), relu=True, name="c1")
            y = b.conv(t, w2.to(DEV), pad=1, scale=s2.to(DEV), shift=h2.to(DEV), relu=True, res="input", name="c2")
        assert [o.kind for o in b.p.ops] == ([f"bb{C}"] if fuse else ["conv", "conv"])
        bufs = dict(b.p._alloc(B, torch.device(DEV)))
        bufs["input"] = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        bufs[y].fill_(float("nan"))
        for op in b.p.ops:
            b.p._launch(lib, op, bufs, B, _lib.current_stream())
        torch.cuda.synchronize()
        outs.append(bufs[y].clone().view(B, H, W, C))
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), int((outs[0] != outs[1]).sum())
    xd = x.double()
    t = torch.relu(torch.nn.functional.conv2d(xd, w1.double(), padding=1) * s1.double().view(1, -1, 1, 1) + h1.double().view(1, -1, 1, 1))
    t = t.bfloat16().double()                                   # the block's intermediate is a bf16 tensor
    ref = torch.relu(torch.nn.functional.conv2d(t, w2.double(), padding=1) * s2.double().view(1, -1, 1, 1) + h2.double().view(1, -1, 1, 1) + xd)
    got = outs[0].float().cpu().permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max() / ref.abs().max() < 8e-3


def test_hrnet_with_fused_basic_blocks_equals_the_per_conv_program_bitwise(golden):
    """HRNet-W32 bf16 end to end with `fuse_blocks` (the 32 + 32 BasicBlocks of the 32- and 64-channel branches as one launch each; the default since
    round 6's strip kernels): same heat maps, bit for bit, as the one-launch-per-conv program."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    g = golden("g3_hrnet_w32_fwd.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = "bf16"
    x = _cuda(synth.input_images(3, 7))
    with torch.no_grad():
        assert m.fuse_blocks                      # (round 6 default)
        m.fuse_blocks = False
        plain = m(x).clone()
        assert sum(op.kind in ("bb32", "bb64") for op in m.hip_program(x).ops) == 0
        m.fuse_blocks = True
        fused = m(x)
        kinds = [op.kind for op in m.hip_program(x).ops]
        assert kinds.count("bb32") == 32 and kinds.count("bb64") == 0
        m.fuse_blocks64 = True                    # (opt-in: the 64-channel branch's blocks too)
        fused64 = m(x)
        kinds = [op.kind for op in m.hip_program(x).ops]
        assert kinds.count("bb32") == 32 and kinds.count("bb64") == 32
        assert torch.equal(plain, fused64)
    assert torch.equal(plain, fused)


def test_hrnet_fused_layer1_equals_the_per_conv_program_bitwise(golden):
    """HRNet-W32 bf16: layer1.0's conv3 + projection shortcut as one launch (`fuse_tail`, sp_dual_pw_bf16) and layer1.1-1.3 as one launch each
    (`fuse_bottlenecks`, sp_bottleneck_c64's eight-wave kernel; round 6 defaults) give the heat maps of the conv-by-conv program bit for bit."""
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    g = golden("g3_hrnet_w32_fwd.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(m.cfg, 17), int(g["seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = "bf16"
    x = _cuda(synth.input_images(3, 7))
    with torch.no_grad():
        m.fuse_tail = m.fuse_bottlenecks = False
        plain = m(x).clone()
        kinds = [op.kind for op in m.hip_program(x).ops]
        assert "dual1x1" not in kinds and "bneck64" not in kinds
        m.fuse_tail = m.fuse_bottlenecks = True
        fused = m(x)
        kinds = [op.kind for op in m.hip_program(x).ops]
    assert kinds.count("dual1x1") == 1 and kinds.count("bneck64") == 3
    assert torch.equal(plain, fused)


# ---------------------------------------------------------------------------------------------- bench.py --gpus N as the driver invokes it
@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_self_launches_two_ranks(mode):
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE): the parent spawns the ranks before touching the GPU and forwards
    rank 0's ONE line.  On a 1-GPU box the two ranks share the device and gloo carries the barrier / reductions."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--dist-backend", backend, "--no-cpu-baseline", "--mode", mode]   # (kernel events on: the roofline passes run on every rank)
    if mode == "train":
        cmd += ["--dtype", "bf16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 16 and line["config"]["images_per_gpu"] == 8
    assert line["value"] > 0 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    # whole-job throughput = units of all ranks / MAX time: consistent with ms_per_step
    assert abs(line["value"] - 16 / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.02
    if mode == "train":
        # an N > 1 train line says which path its collectives took and why (comm_select), and splits the step (round-4 verdict, next 4b)
        chk = line["collective_self_check"]
        if backend == "gloo":
            assert line["collective_path"] == "torch.distributed" and chk["path"] == "torch.distributed" and "gloo" in chk["reason"]
        else:
            assert chk["path"] in ("sp_comm", "torch.distributed") and chk["self_check"] != "not run"      # a peer exists: the check ran
            assert ("sp_comm" in line["collective_path"]) == (chk["path"] == "sp_comm")
            assert "rccl" in line["config"]
        assert "allreduce_wait" in line["step_split_ms"] and line["collectives_per_step"]["gradient_buckets"] >= 1


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_driver_command_runs_the_three_jobs(launcher):
    """The driver's own command at N = 2 (`bench.py --gpus 2 --steps K --warmup W`, headline config untouched) on the GPU: the supervisor runs
    the inference replicas, the collective self-check and the bf16 32-image-per-GPU train step as three N-rank jobs with deadlines, and the
    ONE line carries the train step as `other_configs` with the path its collectives took and why (round-5 verdict, next 1).  On a 1-GPU
    box the ranks share the device over gloo (self-check verdict: not nccl -> torch.distributed); with two GPUs it is RCCL and the
    comparison really runs.  `torchrun`: the way the driver starts N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) -
    every worker then supervises the one child of its rank and the supervisors meet through files (simple_pose_amd/launch.py)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SP_NATIVE_COMM", "SP_BENCH_CHILD")}
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--dist-backend", backend, "--no-kernel-events"]
    if launcher == "torchrun":
        from simple_pose_amd.launch import free_port
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + cmd
    else:
        cmd = [sys.executable] + cmd
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["dtype"] == "f32" and line["value"] > 0
    assert list(line)[-1] == "other_configs" and len(line["other_configs"]) == 1
    oc = line["other_configs"][0]
    assert "error" not in oc, oc
    assert oc["job"]["status"] == "ok" and oc["n_gpus"] == 2 and oc["global_batch"] == 64 and oc["dtype"] == "bf16" and oc["value"] > 0
    assert abs(oc["value"] - 64 / (oc["ms_per_step"] * 1e-3)) / oc["value"] < 0.02
    chk = oc["collective_self_check"]
    assert chk["job"]["status"] == "ok"
    if backend == "gloo":
        assert oc["collective_path"] == "torch.distributed" and "gloo" in chk["reason"] and chk["self_check"] == "not run"
    else:
        assert chk["self_check"] != "not run" and ("sp_comm" in oc["collective_path"]) == chk["native"] and oc["rccl"] is not None
    assert "allreduce_wait" in oc["step_split_ms"] and oc["collectives_per_step"]["gradient_buckets"] >= 1


def test_bench_micro_mode_prints_one_json_line_on_stdout():
    """`python bench.py --mode micro`: the summary is the ONE line on the real stdout (bench.py redirects file descriptor 1 to stderr before
    any library can print there, so tools/bench_micro.py hands its summary to bench.py's `emit`; round-4 advisor finding)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "micro"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"].startswith("HBM-bound kernels") and d["rows"] > 10


@pytest.mark.gpu
@pytest.mark.parametrize("knob,selector", [("SP_BB32_W8", "test_fused_basic_block_c32_is_bit_identical_to_its_two_convs and 5-33-47-32"),
                                           ("SP_BNECK_W8", "test_fused_bottlenecks_equal_the_per_conv_program_bitwise and duc")])
def test_kernels_kept_for_same_box_ab_still_match(knob, selector):
    """The round-2 four-wave BasicBlock kernel (SP_BB32_W8=0) and the four-wave fused bottleneck (SP_BNECK_W8=0) stay in the library for same-box A/Bs
    (profiles/r06_bb32_ab.txt, r06_bneck8_ab.txt): the knobs are read once per process, so a fresh interpreter runs one parity case of each on them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{knob: "0"})
    selector += " and not test_kernels_kept_for_same_box_ab"           # (this test's own id contains the selector: never select it in the child)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", selector],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
