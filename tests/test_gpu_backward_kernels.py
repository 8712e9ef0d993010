"""Per-kernel parity of the BACKWARD half of the train step, each entry point of the C ABI on its own against
float64 torch on well-conditioned random operands - the whole-net comparisons of test_gpu_train.py are chaos-limited (B=2 BatchNorm
net: bars of 2e-2) and would hide a 1e-3 error in one layer.

Reference semantics: loss.backward() through nn.Conv2d / nn.ConvTranspose2d(4,2,1) / nn.BatchNorm2d / nn.ReLU
(nets/pose_resnet_dconv.py:99-103,158,236-244) and torch.optim.Adam (processors/ddp_pose_resnet_solver.py:70-72,117-119).

Every bar below is pinned to what was measured on the MI355X (`measured` fixture -> gpurun_out/measured_parity.json), never looser
than 3x the measurement.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from simple_pose_amd import _lib, synth  # noqa: E402
from simple_pose_amd.train import ConvT, FlatParams  # noqa: E402

DEV = "cuda:0"
P = _lib.ptr


class _OneLayer:
    """The slice of PoseTrainer a ConvT needs: flat parameter / gradient buffers of ONE layer, the wgrad workspace, and the layer's
    pack jobs run one by one through sp_permute4_f32 (PoseTrainer.repack batches the same jobs through sp_permute4_batched)."""
    kernel_events = None
    _wgrad_stream = None

    def __init__(self, kind, weight, h, w, bf16, g16=False, **kw):
        self.bf16 = bf16
        self.g16 = g16
        self.grad_dtype = torch.bfloat16 if g16 else torch.float32
        mod = torch.nn.Module()
        mod.c = torch.nn.Module()
        mod.c.weight = torch.nn.Parameter(weight.to(DEV))
        self.flat = FlatParams(mod)
        self.wgrad_ws = torch.empty(48 * 1024 * 1024, dtype=torch.float32, device=DEV)
        self.layer = ConvT(self, "c", kind, mod.c.weight.detach(), h, w, **kw)
        lib, st = _lib.lib(), _lib.current_stream()
        for j in self.layer.pack_jobs:
            o, _ = self.flat.offsets[j.src_name]
            _lib.check(lib.sp_permute4_f32(P(self.flat.data), P(j.dst), int(j.dst.dtype == torch.bfloat16), (ctypes.c_int32 * 4)(*j.dims),
                                           (ctypes.c_int64 * 4)(*j.strides), (ctypes.c_int32 * 4)(*j.valid), o + j.base, j.dst_off, st), "pack")


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.pow(2).mean().sqrt())


def _rel_bf16(got, ref):
    """for a bf16 result: max |got - ref| / max(|ref|, rms(ref)) - one rounding to 8 significant bits is 2^-9 = 1.95e-3 of the value"""
    got, ref = got.double().cpu(), ref.double().cpu()
    return float(((got - ref).abs() / torch.maximum(ref.abs(), ref.pow(2).mean().sqrt())).max())


def _bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _nhwc(t, c_buf=None, dtype=torch.float32):
    """NCHW float64 host tensor -> NHWC device tensor of `dtype`, channels zero-padded to c_buf."""
    t = t.permute(0, 2, 3, 1)
    if c_buf is not None and c_buf > t.shape[-1]:
        t = F.pad(t, (0, c_buf - t.shape[-1]))
    return t.contiguous().to(dtype).to(DEV)


# name, kind, I, O, k, stride, pad, H, W, B
LAYERS = [
    ("1x1", "conv", 64, 128, 1, 1, 0, 12, 10, 3),
    ("1x1_wide_m", "conv", 64, 256, 1, 1, 0, 64, 48, 4),           # M = 12,288 pixels: many pixel splits per dW tile
    ("1x1_deep_k", "conv", 1024, 256, 1, 1, 0, 4, 3, 5),           # M = 60: fewer pixels than one split
    ("3x3_s1", "conv", 64, 64, 3, 1, 1, 9, 7, 2),
    ("3x3_s1_k4608", "conv", 512, 128, 3, 1, 1, 8, 6, 2),
    ("3x3_s2", "conv", 64, 128, 3, 2, 1, 12, 8, 2),
    ("1x1_s2_shortcut", "conv", 64, 128, 1, 2, 0, 12, 8, 2),       # dgrad reaches phase (0,0) only
    ("final_1x1_17", "conv", 64, 17, 1, 1, 0, 16, 12, 2),          # 17 heat-map channels in a K-tile-padded gradient buffer
    ("final_3x3_17", "conv", 128, 17, 3, 1, 1, 16, 12, 2),         # the DUC head's final layer
    ("deconv_k4s2p1", "deconv", 128, 64, 4, 2, 1, 6, 5, 2),
    ("deconv_k4s2p1_2048", "deconv", 2048, 256, 4, 2, 1, 4, 3, 2),
]
EPS32 = 2.0 ** -24
# Error = max |got - ref| over rms(ref).  Measured on the MI355X (round 3):
#   wgrad (reduction over the M pixels, cut into ranges of ~800 (fp32) / ~1,700 (bf16) pixels that are summed in order by one workgroup
#   each and folded in a fixed order): one fp32 accumulation chain per range, so the error grows like sqrt(min(M, range)):
#   fp32 <= 2.1e-6 up to M = 400, 5.9e-6 at M = 12,288 (0.9 x EPS32 sqrt(M)); bf16 operands (exact products, fp32 accumulation;
#   reference = float64 on the same bf16-rounded operands, dW stays fp32) <= 6.7e-7, 1.4e-6 at M = 12,288;
#   dgrad = one fp32 accumulation chain of length K per output (v_mfma_f32_32x32x2_f32 == an fmaf chain in k order): the error grows
#   like sqrt(K): measured 2.5-3.5 x EPS32 sqrt(K) in fp32 (1.7e-6 at K = 64 ... 1.3e-5 at K = 4096), 0.8-1.0 x EPS32 sqrt(K) in bf16.
def wgrad_bar(bf16, M):
    return max(1.5e-6, 0.5 * EPS32 * float(np.sqrt(M))) if bf16 else max(4e-6, 2.0 * EPS32 * float(np.sqrt(M)))
DGRAD_BAR_K = {False: 6.0, True: 2.0}        # x EPS32 * sqrt(K)


def _reference(kind, I, O, k, s, p, H, W, B, bf16, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, I, H, W, generator=g, dtype=torch.float64)
    w = torch.randn((O, I, k, k) if kind == "conv" else (I, O, k, k), generator=g, dtype=torch.float64) / np.sqrt(I * k * k / (4 if kind == "deconv" else 1))
    w = w.float().double()                      # the layer's master weight is fp32
    if bf16:
        x = _bf(x)
    wq = _bf(w) if bf16 else w                   # packed copies are bf16 in bf16 mode
    xr, wr = x.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=s, padding=p) if kind == "conv" else F.conv_transpose2d(xr, wr, stride=2, padding=1)
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    if bf16:
        dz = _bf(dz)
    y.backward(dz)
    return x, w, dz, xr.grad, wr.grad


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
def test_wgrad_and_dgrad_vs_float64(case, bf16, measured):
    name, kind, I, O, k, s, p, H, W, B = case
    x, w, dz, dx_ref, dw_ref = _reference(kind, I, O, k, s, p, H, W, B, bf16, seed=11)
    one = _OneLayer(kind, w.float(), H, W, bf16, stride=s, pad=p)
    L = one.layer
    lib, st = _lib.lib(), _lib.current_stream()
    adt = torch.bfloat16 if bf16 else torch.float32
    xd = _nhwc(x, dtype=adt)
    dzd = _nhwc(dz, c_buf=L.c_out_buf, dtype=adt)
    # ---- wgrad: written straight into the reference weight layout inside the flat gradient buffer ----
    L.d_wgrad.batch = B
    gt, at = (dzd, xd) if kind == "conv" else (xd, dzd)
    one.flat.grad.fill_(float("nan"))            # every element of the layer's gradient must be written
    _lib.check(lib.sp_conv2d_wgrad(L.d_wgrad, P(gt), gt.shape[-1], P(at), L.wg["n_valid"], L.wg["c_valid"], L.wg["kw_valid"], L.wg["s_n"],
                                   L.wg["s_c"], P(one.flat.view("c.weight", grad=True)), P(one.wgrad_ws), one.wgrad_ws.numel() * 4, st), name)
    torch.cuda.synchronize()
    dw = one.flat.view("c.weight", grad=True).view(w.shape)
    assert torch.isfinite(dw).all()
    e = _rel(dw, dw_ref)
    wbar = wgrad_bar(bf16, L.d_wgrad.batch * L.d_wgrad.grid_h * L.d_wgrad.grid_w)
    measured("wgrad_rel", e, wbar)
    assert e <= wbar
    # a second launch reproduces the bits (fixed reduction order, no float atomics)
    first = dw.clone()
    _lib.check(lib.sp_conv2d_wgrad(L.d_wgrad, P(gt), gt.shape[-1], P(at), L.wg["n_valid"], L.wg["c_valid"], L.wg["kw_valid"], L.wg["s_n"],
                                   L.wg["s_c"], P(one.flat.view("c.weight", grad=True)), P(one.wgrad_ws), one.wgrad_ws.numel() * 4, st), name)
    torch.cuda.synchronize()
    assert torch.equal(first, dw)
    # ---- dgrad: the forward kernel on the re-packed weights (flipped taps / stride-2 phases / deconv as a 4x4 stride-2 conv) ----
    full = L.dgrad_full_cover
    dxd = (torch.empty if full else torch.zeros)((B, H, W, I), dtype=torch.float32, device=DEV)
    for d, wd in zip(L.d_dgrad, L.w_dgrad):
        d.batch = B
        _lib.check(lib.sp_conv2d_fwd(d, P(dzd), P(wd), None, None, None, P(dxd), st), name + ".dgrad")
    torch.cuda.synchronize()
    dbar = DGRAD_BAR_K[bf16] * EPS32 * float(np.sqrt(max(d.k_pad for d in L.d_dgrad)))
    e = _rel(dxd.permute(0, 3, 1, 2), dx_ref)
    measured("dgrad_rel", e, dbar)
    assert e <= dbar
    # accumulate form (residual fan-out): dx = acc + dgrad, in place
    if full:
        acc = torch.randn(B, H, W, I, generator=torch.Generator().manual_seed(3)).to(DEV)
        acc0 = acc.clone()
        for d, wd in zip(L.d_dgrad, L.w_dgrad):
            _lib.check(lib.sp_conv2d_fwd(d, P(dzd), P(wd), None, None, P(acc), P(acc), st), name + ".dgrad+")
        torch.cuda.synchronize()
        e = _rel(acc.permute(0, 3, 1, 2), dx_ref + acc0.cpu().double().permute(0, 3, 1, 2))
        measured("dgrad_accumulate_rel", e, dbar)
        assert e <= dbar


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_batched_wgrad_equals_layer_by_layer_bitwise(bf16):
    """sp_conv2d_wgrad_batched over every test layer at once (mixed dW tile shapes, more layers than one launch's table holds) leaves
    the same bits as one sp_conv2d_wgrad call per layer: how a layer's pixels are cut depends on the layer alone."""
    lib, st = _lib.lib(), _lib.current_stream()
    adt = torch.bfloat16 if bf16 else torch.float32
    keep, single = [], []
    cases = LAYERS + [(n + "_again", *rest) for n, *rest in LAYERS[:8]]          # 19 jobs: two unit launches
    jobs = (_lib.WgradJob * len(cases))()
    for job, (name, kind, I, O, k, s, p, H, W, B) in zip(jobs, cases):
        x, w, dz, _, _ = _reference(kind, I, O, k, s, p, H, W, B, bf16, seed=len(keep))
        one = _OneLayer(kind, w.float(), H, W, bf16, stride=s, pad=p)
        L = one.layer
        xd, dzd = _nhwc(x, dtype=adt), _nhwc(dz, c_buf=L.c_out_buf, dtype=adt)
        L.wgrad(xd, dzd, B)
        torch.cuda.synchronize()
        single.append(one.flat.view("c.weight", grad=True).clone())
        one.flat.grad.fill_(float("nan"))
        L.wgrad_job(xd, dzd, B, job)
        keep.append((one, xd, dzd))
    need = ctypes.c_int64(0)
    _lib.check(lib.sp_conv2d_wgrad_workspace(jobs, len(cases), ctypes.byref(need)))
    ws = torch.empty(need.value // 4 + 16, dtype=torch.float32, device=DEV)
    assert lib.sp_conv2d_wgrad_batched(jobs, len(cases), P(ws), need.value - 4, st) != 0          # a short workspace is refused
    assert b"workspace too small" in lib.sp_last_error()
    _lib.check(lib.sp_conv2d_wgrad_batched(jobs, len(cases), P(ws), need.value, st), "batched")
    torch.cuda.synchronize()
    for (one, _, _), ref, case in zip(keep, single, cases):
        assert torch.equal(one.flat.view("c.weight", grad=True), ref), case[0]


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_stem_7x7_wgrad_vs_float64(bf16, measured):
    """conv1 (3 -> 64, 7x7 s2 p3) reads the image as NHWC4 (fp32) / NHWC8 (bf16) with 8 packed taps per row; only its weight
    gradient exists (the input needs none)."""
    I, O, k, s, p, H, W, B = 3, 64, 7, 2, 3, 32, 24, 3
    x, w, dz, _, dw_ref = _reference("conv", I, O, k, s, p, H, W, B, bf16, seed=5)
    cbuf = 8 if bf16 else 4
    one = _OneLayer("conv", w.float(), H, W, bf16, stride=s, pad=p, c_in_buf=cbuf, need_dgrad=False)
    L = one.layer
    adt = torch.bfloat16 if bf16 else torch.float32
    xd, dzd = _nhwc(x, c_buf=cbuf, dtype=adt), _nhwc(dz, dtype=adt)
    L.d_wgrad.batch = B
    one.flat.grad.fill_(float("nan"))
    _lib.check(_lib.lib().sp_conv2d_wgrad(L.d_wgrad, P(dzd), dzd.shape[-1], P(xd), L.wg["n_valid"], L.wg["c_valid"], L.wg["kw_valid"],
                                          L.wg["s_n"], L.wg["s_c"], P(one.flat.view("c.weight", grad=True)), P(one.wgrad_ws),
                                          one.wgrad_ws.numel() * 4, _lib.current_stream()), "stem")
    torch.cuda.synchronize()
    dw = one.flat.view("c.weight", grad=True).view(w.shape)
    assert torch.isfinite(dw).all()
    e = _rel(dw, dw_ref)
    wbar = wgrad_bar(bf16, B * L.d_wgrad.grid_h * L.d_wgrad.grid_w)
    measured("wgrad_rel", e, wbar)
    assert e <= wbar


# measured (round 3): fp32: dz <= 1.6e-6 of its rms, dgamma / dbeta <= 3.8e-7, dres <= 6.8e-7; mean <= 1.3e-7, invstd <= 1.6e-7.  bf16
# activations: y and dz are stored as bf16 - one rounding, 2^-9 of the value (_rel_bf16); the sums stay fp32
BN_BAR = {False: dict(dz=3e-6, sums=1e-6, stats=4e-7, dres=1.5e-6), True: dict(dz=4e-3, sums=1e-6, stats=4e-7, dres=1.5e-6)}


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows_hw,C,relu,res_on", [((4, 16, 12), 64, True, False), ((3, 9, 7), 256, True, True), ((2, 5, 3), 2048, False, False),
                                                   ((8, 32, 24), 128, True, True)])
def test_batchnorm_train_backward_vs_float64(rows_hw, C, relu, res_on, bf16, measured):
    """y = [relu](bn_train(z) [+ res]) -> (dz, dgamma, dbeta, dres) through sp_bn_train_bwd_nhwc and through its two halves
    sp_bn_train_bwd_reduce_nhwc + sp_bn_train_bwd_apply_nhwc (the SyncBatchNorm form), against float64 autograd."""
    B, H, W = rows_hw
    rows = B * H * W
    g = torch.Generator().manual_seed(C + rows)
    z = torch.randn(rows, C, generator=g, dtype=torch.float64) * (0.5 + torch.rand(C, generator=g, dtype=torch.float64)) + torch.randn(C, generator=g, dtype=torch.float64)
    res = torch.randn(rows, C, generator=g, dtype=torch.float64) if res_on else None
    gamma = (0.75 + 0.5 * torch.rand(C, generator=g, dtype=torch.float64)).float().double()
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).float().double()
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64).float().double()
    adt = torch.bfloat16 if bf16 else torch.float32
    if bf16:
        z = _bf(z)
        res = _bf(res) if res_on else None
    lib, st = _lib.lib(), _lib.current_stream()
    ws = torch.empty(4 << 20, dtype=torch.uint8, device=DEV)
    zd, dyd = z.to(adt).to(DEV), dy.float().to(DEV)
    resd = res.to(adt).to(DEV) if res_on else None
    gd, bd = gamma.float().to(DEV), beta.float().to(DEV)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    _lib.check(lib.sp_bn_train_stats_nhwc(P(zd), int(bf16), rows, C, 1e-5, 0.1, P(mean), P(invstd), P(rm), P(rv), P(ws), st), "stats")
    yd = torch.empty((rows, C), dtype=adt, device=DEV)
    _lib.check(lib.sp_bn_apply_nhwc(P(zd), int(bf16), P(mean), P(invstd), P(gd), P(bd), P(resd), P(yd), rows, C, int(relu), None, st), "apply")
    torch.cuda.synchronize()
    # float64 reference; the ReLU mask is taken from the kernel's own y (a pre-activation within rounding of 0 has no defined sign)
    zr = z.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if res_on else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    m64, v64 = zr.mean(0), zr.var(0, unbiased=False)
    pre = (zr - m64) / torch.sqrt(v64 + 1e-5) * gr + br
    if res_on:
        pre = pre + rr
    mask = (yd.double().cpu() > 0).double() if relu else torch.ones_like(pre)
    (pre * mask).backward(dy)
    measured("mean_rel", _rel(mean, m64.detach()), BN_BAR[bf16]["stats"])
    measured("invstd_rel", _rel(invstd, 1 / torch.sqrt(v64.detach() + 1e-5)), BN_BAR[bf16]["stats"])
    assert _rel(mean, m64.detach()) <= BN_BAR[bf16]["stats"] and _rel(invstd, 1 / torch.sqrt(v64.detach() + 1e-5)) <= BN_BAR[bf16]["stats"]
    assert _rel(rm, 0.1 * m64.detach()) <= 1e-6 and _rel(rv, 0.9 + 0.1 * zr.detach().var(0, unbiased=True)) <= 1e-6
    y_ref = torch.relu(pre.detach()) if relu else pre.detach()
    e = (_rel_bf16 if bf16 else _rel)(yd.float(), y_ref)
    measured("y_rel", e, 4e-3 if bf16 else 2e-6)
    assert e <= (4e-3 if bf16 else 2e-6)

    def check(tag, dz, dgamma, dbeta, dres):
        for nm, got, ref, bar in (("dz", dz.float(), zr.grad, BN_BAR[bf16]["dz"]), ("dgamma", dgamma, gr.grad, BN_BAR[bf16]["sums"]),
                                  ("dbeta", dbeta, br.grad, BN_BAR[bf16]["sums"])) + ((("dres", dres, rr.grad, BN_BAR[bf16]["dres"]),) if res_on else ()):
            e = (_rel_bf16 if (bf16 and nm == "dz") else _rel)(got, ref)
            measured(f"{tag}/{nm}_rel", e, bar)
            assert e <= bar, (tag, nm, e)

    rs = P(yd) if relu else None
    dz = torch.empty((rows, C), dtype=adt, device=DEV)
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dres = torch.empty((rows, C), device=DEV) if res_on else None
    _lib.check(lib.sp_bn_train_bwd_nhwc(P(dyd), int(bf16), rs, P(zd), P(mean), P(invstd), P(gd), rows, C, P(dz), P(dgam), P(dbet), P(dres), 0,
                                        P(ws), st), "bwd")
    torch.cuda.synchronize()
    check("one_call", dz, dgam, dbet, dres)
    # the two halves, dres accumulated onto an existing gradient
    dz2 = torch.empty_like(dz)
    dgam2, dbet2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    base = torch.randn(rows, C, generator=torch.Generator().manual_seed(1)).to(DEV) if res_on else None
    dres2 = base.clone() if res_on else None
    _lib.check(lib.sp_bn_train_bwd_reduce_nhwc(P(dyd), int(bf16), rs, P(zd), P(mean), P(invstd), rows, C, P(dgam2), P(dbet2), P(ws), st), "reduce")
    _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(dyd), int(bf16), rs, P(zd), P(mean), P(invstd), P(gd), P(dgam2), P(dbet2), rows, rows, C, P(dz2),
                                              P(dres2), 1, st), "apply")
    torch.cuda.synchronize()
    check("two_halves", dz2, dgam2, dbet2, (dres2 - base) if res_on else None)
    assert torch.equal(dz, dz2) and torch.equal(dgam, dgam2) and torch.equal(dbet, dbet2)


def _ulp_diff(a: torch.Tensor, b: torch.Tensor, *operands) -> float:
    """max |a - b| in units of the fp32 spacing at the largest magnitude among b and the operands it was formed from (a sum that
    cancels has no meaningful ulp of its own; a parameter far smaller than its update is as accurate as the update)"""
    a, b = a.double().cpu(), b.double().cpu()
    mag = b.abs()
    for o in operands:
        mag = torch.maximum(mag, o.double().cpu().abs() if isinstance(o, torch.Tensor) else torch.full_like(mag, abs(o)))
    spacing = torch.from_numpy(np.spacing(mag.float().numpy())).double()
    return float(((a - b).abs() / spacing).max())


def test_adam_ten_steps_vs_torch_optim(measured):
    """sp_adam_step against torch.optim.Adam (lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay; ddp...:70-72): ten steps with fresh
    gradients, (a) per step from the same state, (b) free running.  Unit: fp32 spacing at the largest operand of each result (the
    moments' two terms; max(|p|, lr) for a parameter) - torch 2.x forms exp_avg with lerp_, the reference's torch >= 1.5 with mul_ / add_,
    the kernel with one fused multiply-add: the same value to an ulp of the larger term."""
    n = 1 << 16
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g))) for _ in range(10)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    lib, st = _lib.lib(), _lib.current_stream()
    free_p, free_m, free_v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    worst_step = worst_free = worst_state = 0.0
    for t, gt in enumerate(grads, start=1):
        # (a) one step from torch's state before the step
        state = opt.state[ref] if t > 1 else None
        sp = ref.detach().clone().to(DEV)
        sm = state["exp_avg"].clone().to(DEV) if state else torch.zeros(n, device=DEV)
        sv = state["exp_avg_sq"].clone().to(DEV) if state else torch.zeros(n, device=DEV)
        gd = gt.to(DEV)
        m_before, v_before = sm.cpu(), sv.cpu()
        _lib.check(lib.sp_adam_step(P(sp), P(gd), P(sm), P(sv), n, 1e-3, 0.9, 0.999, 1e-8, t, 1.0, st), "adam")
        _lib.check(lib.sp_adam_step(P(free_p), P(gd), P(free_m), P(free_v), n, 1e-3, 0.9, 0.999, 1e-8, t, 1.0, st), "adam")
        ref.grad = gt.clone()
        opt.step()
        torch.cuda.synchronize()
        worst_step = max(worst_step, _ulp_diff(sp, ref.detach(), 1e-3))
        worst_state = max(worst_state, _ulp_diff(sm, opt.state[ref]["exp_avg"], m_before, 0.1 * gt),
                          _ulp_diff(sv, opt.state[ref]["exp_avg_sq"], v_before, 1e-3 * gt * gt))
        worst_free = max(worst_free, _ulp_diff(free_p, ref.detach(), 1e-3))
    measured("param_ulp_per_step", worst_step, 2)
    measured("moment_ulp_per_step", worst_state, 2)
    measured("param_ulp_free_running_10_steps", worst_free, 25)        # (different moments after a few steps: measured 12.5)
    assert worst_step <= 2 and worst_state <= 2 and worst_free <= 25
    # grad_scale (1 / world size after a SUM all-reduce) == scaling the gradient first
    a_p, a_m, a_v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    b_p, b_m, b_v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    g8 = (grads[0] * 8).to(DEV)
    _lib.check(lib.sp_adam_step(P(a_p), P(g8), P(a_m), P(a_v), n, 1e-3, 0.9, 0.999, 1e-8, 1, 0.125, st), "adam")
    _lib.check(lib.sp_adam_step(P(b_p), P(grads[0].to(DEV)), P(b_m), P(b_v), n, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, st), "adam")
    torch.cuda.synchronize()
    assert torch.equal(a_p, b_p) and torch.equal(a_m, b_m) and torch.equal(a_v, b_v)


@pytest.mark.parametrize("B,J,hw", [(32, 17, 64 * 48), (3, 5, 35), (1, 17, 4), (130, 2, 9000)])
def test_bias_gradient_channel_sum_nchw_vs_float64(B, J, hw, measured):
    """sp_channel_sum_nchw (final_layer.bias.grad = d loss / d heat maps summed over batch and pixels, ddp...:117-118): accumulated in
    fp64 on the device, so the result is the correctly rounded fp32 of the exact sum (both the float4 walk and the any-size fallback);
    a view at an odd float offset takes the fallback too."""
    g = torch.Generator().manual_seed(B * 1000 + hw)
    x = (torch.randn((B, J, hw), generator=g) * 3.0).to(DEV)
    lib, st = _lib.lib(), _lib.current_stream()
    worst = 0.0
    for off in (0, 1):                                      # off = 1: a start that is only 4-byte aligned (scalar walk)
        buf = torch.empty(B * J * hw + 1, device=DEV)
        src = buf[off:off + B * J * hw].view(B, J, hw)
        src.copy_(x)
        out = torch.full((J,), float("nan"), device=DEV)
        _lib.check(lib.sp_channel_sum_nchw(P(src), B, J, hw, P(out), st), "bias.grad")
        torch.cuda.synchronize()
        ref = src.double().sum(dim=(0, 2))
        worst = max(worst, float(((out.double() - ref).abs() / torch.from_numpy(np.spacing(ref.float().abs().cpu().numpy())).double().to(DEV)).max()))
    measured("ulp_of_exact_sum", worst, 0.51)
    assert worst <= 0.51


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("B,HW,C", [(3, 48, 256), (2, 192, 512), (5, 7, 64)])
def test_selayer_gate_backward_vs_float64(B, HW, C, bf16, measured):
    """The four kernels of the SELayer backward (sp_se_gate_bwd_reduce / sp_se_sigmoid_bwd / sp_relu_bwd_rows / sp_se_gate_bwd_apply) against
    torch autograd in float64 on y = relu(u * sigmoid(g) + identity) and h = relu(.) with the saved activations rounded as the forward
    stores them (bf16 mode); gradients w.r.t. activations are fp32."""
    g_ = torch.Generator().manual_seed(B * 100 + C)
    adt = torch.bfloat16 if bf16 else torch.float32
    u = torch.randn((B, HW, C), generator=g_).to(adt)
    idn = torch.randn((B, HW, C), generator=g_).to(adt)
    gl = (torch.randn((B, C), generator=g_) * 1.5).to(adt)
    dy = torch.randn((B, HW, C), generator=g_)
    ds = torch.randn((B, C), generator=g_)
    u64, i64, g64 = (t.double().requires_grad_(True) for t in (u, idn, gl))
    a64 = torch.sigmoid(g64)
    y64 = torch.relu(u64 * a64.view(B, 1, C) + i64)
    y = y64.detach().to(adt)                                          # what the forward kernel stores
    dr = torch.where(y.double() > 0, dy.double(), torch.zeros_like(dy.double()))
    da_ref = (dr * u.double()).sum(1)
    a = torch.sigmoid(gl.double())
    dg_ref = da_ref * a * (1 - a)
    du_ref = dr * a.view(B, 1, C) + ds.double().view(B, 1, C) / HW
    lib, st = _lib.lib(), _lib.current_stream()
    ud, yd, gd, dyd, dsd = (t.to(DEV).contiguous() for t in (u, y, gl, dy, ds))
    da = torch.empty((B, C), device=DEV)
    _lib.check(lib.sp_se_gate_bwd_reduce(P(dyd), int(bf16), P(yd), P(ud), B, HW, C, P(da), st), "reduce")
    dg = torch.empty((B, C), dtype=adt, device=DEV)
    db = torch.empty(C, device=DEV)
    _lib.check(lib.sp_se_sigmoid_bwd(P(da), int(bf16), P(gd), B, C, P(dg), P(db), st), "sigmoid")
    du = torch.empty((B, HW, C), device=DEV)
    base = torch.randn((B, HW, C), generator=g_).to(DEV)
    dres = base.clone()
    _lib.check(lib.sp_se_gate_bwd_apply(P(dyd), int(bf16), P(yd), P(gd), P(dsd), B, HW, C, P(du), P(dres), 1, st), "apply")
    dres0 = torch.empty_like(dres)
    _lib.check(lib.sp_se_gate_bwd_apply(P(dyd), int(bf16), P(yd), P(gd), P(dsd), B, HW, C, P(du), P(dres0), 0, st), "apply")
    torch.cuda.synchronize()
    e_da = _rel(da.cpu().double(), da_ref)
    e_dg = (_rel_bf16 if bf16 else _rel)(dg.float().cpu().double(), dg_ref)      # bf16: one rounding of the stored operand
    e_du = _rel(du.cpu().double(), du_ref)
    measured("da_rel", e_da, 1e-6)
    measured("dg_rel", e_dg, 4e-3 if bf16 else 2e-6)
    measured("du_rel", e_du, 2e-6)
    assert e_da < 1e-6 and e_dg < (4e-3 if bf16 else 2e-6) and e_du < 2e-6
    assert torch.equal(dres0.cpu(), dr.float()) and torch.allclose(dres.cpu(), base.cpu() + dr.float(), rtol=0, atol=1e-6)
    assert _rel(db.cpu().double(), dg_ref.sum(0)) < 2e-6             # the bias gradient sums the UNROUNDED dg (fp64 accumulation)
    # relu rows
    h = torch.randn((B, C), generator=g_).to(adt)
    dh = torch.randn((B, C), generator=g_)
    out = torch.empty((B, C), dtype=adt, device=DEV)
    db1 = torch.empty(C, device=DEV)
    dhd, hd = dh.to(DEV), h.to(DEV)
    _lib.check(lib.sp_relu_bwd_rows(P(dhd), int(bf16), P(hd), B, C, P(out), P(db1), st), "relu")
    torch.cuda.synchronize()
    ref = torch.where(h.float() > 0, dh, torch.zeros_like(dh))
    assert torch.equal(out.cpu(), ref.to(adt)) and _rel(db1.cpu().double(), ref.double().sum(0)) < 1e-6


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("B,h,w,C,f,relu", [(2, 4, 3, 32, 2, True), (1, 2, 2, 64, 8, False), (3, 8, 6, 32, 1, True), (2, 3, 5, 128, 4, True)])
def test_upsample_add_backward_vs_float64(B, h, w, C, f, relu, bf16):
    """sp_upsample_add_bwd_nhwc (HRNet fuse layers, nets/pose_hrnet.py:192-202,250-257) against torch autograd through
    relu(base + interpolate(x, nearest)); both accumulate flags."""
    g_ = torch.Generator().manual_seed(h * 100 + f)
    adt = torch.bfloat16 if bf16 else torch.float32
    x = torch.randn((B, C, h, w), generator=g_).double().requires_grad_(True)
    base = torch.randn((B, C, h * f, w * f), generator=g_).double().requires_grad_(True)
    y = base + F.interpolate(x, scale_factor=f, mode="nearest")
    y = torch.relu(y) if relu else y
    dy = torch.randn(y.shape, generator=g_)
    yq = y.detach().to(adt)                                         # the mask comes from the stored activation
    # (re-derive the reference with the stored mask so that a value rounding to 0 in bf16 is treated alike)
    dr = torch.where(yq.double() > 0, dy.double(), torch.zeros_like(dy.double())) if relu else dy.double()
    dx_ref = F.avg_pool2d(dr, f) * (f * f)
    nhwc = lambda t_: t_.permute(0, 2, 3, 1).contiguous()
    lib, st = _lib.lib(), _lib.current_stream()
    dyd, yd = nhwc(dy).to(DEV), nhwc(yq).to(DEV)
    prev_b, prev_x = torch.randn((B, h * f, w * f, C), generator=g_).to(DEV), torch.randn((B, h, w, C), generator=g_).to(DEV)
    for acc in (0, 1):
        db = prev_b.clone() if acc else torch.full((B, h * f, w * f, C), float("nan"), device=DEV)
        dx = prev_x.clone() if acc else torch.full((B, h, w, C), float("nan"), device=DEV)
        _lib.check(lib.sp_upsample_add_bwd_nhwc(P(dyd), int(bf16), P(yd) if relu else None, B, h, w, C, f, P(db), acc, P(dx), acc, st), "bwd")
        torch.cuda.synchronize()
        ref_b = nhwc(dr).float() + (prev_b.cpu() if acc else 0)
        ref_x = nhwc(dx_ref) + (prev_x.cpu().double() if acc else 0)
        assert torch.allclose(db.cpu(), ref_b, rtol=0, atol=1e-6)
        assert float((dx.cpu().double() - ref_x).abs().max()) <= 1e-5 * max(1.0, float(ref_x.abs().max()))


# ---- round 4: BatchNorm sums out of the conv epilogues, folded by a stand-alone launch or in the prologue of the consuming pass ----------
# (sp_conv2d_fwd_bn_stats -> sp_bn_train_stats_from_conv + sp_bn_apply_nhwc  ==  sp_bn_fold_apply_nhwc, bit for bit;
#  sp_conv2d_dgrad_bn_bwd_stats[2] -> sp_bn_bwd_sums_from_conv + sp_bn_train_bwd_apply_nhwc  ==  sp_bn_fold_bwd_apply_nhwc, bit for bit;
#  both against float64 on the tensors the launches themselves stored.)  Reference: nn.BatchNorm2d in train mode behind nn.Conv2d,
#  nets/pose_resnet_dconv.py:112-133, and its backward.
STATS_CASES = [
    # name, I, O, k, stride, pad, H, W, B
    ("1x1_64_128", 64, 128, 1, 1, 0, 16, 12, 4),
    ("3x3_64_64", 64, 64, 3, 1, 1, 9, 7, 2),           # ragged last M tile
    ("1x1_256_1024", 256, 1024, 1, 1, 0, 8, 6, 3),
    ("3x3_s2_128_128", 128, 128, 3, 2, 1, 16, 12, 2),  # stride-2: dgrad = 4 phase launches filling consecutive partial rows
    ("1x1_wide_m", 64, 256, 1, 1, 0, 64, 48, 2),       # 6,144 pixels: 48-96 partial rows per slab
]


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_conv_epilogue_statistics_and_fused_fold_forward(case, bf16, measured):
    name, I, O, k, s, p, H, W, B = case
    g = torch.Generator().manual_seed(5 + I + O)
    x = torch.randn(B, I, H, W, generator=g, dtype=torch.float64)
    w = (torch.randn(O, I, k, k, generator=g, dtype=torch.float64) / np.sqrt(I * k * k)).float()
    adt = torch.bfloat16 if bf16 else torch.float32
    one = _OneLayer("conv", w, H, W, bf16, stride=s, pad=p)
    L = one.layer
    xd = _nhwc(x, dtype=adt)
    z, part, prow = L.forward_bn_stats(xd, B)
    rows, C = z.shape[0] * z.shape[1] * z.shape[2], O
    torch.cuda.synchronize()
    assert part.shape[1] == prow and prow > 0
    z64 = z.double().cpu().reshape(rows, C)
    # the partial rows add up to the column sums of the STORED tensor
    s0, s1 = part[0].double().sum(0).cpu()[:C], part[1].double().sum(0).cpu()[:C]
    e0 = float((s0 - z64.sum(0)).abs().max() / z64.abs().sum(0).max())
    e1 = float((s1 - (z64 * z64).sum(0)).abs().max() / (z64 * z64).sum(0).max())
    measured("partial_rows_sum_rel", e0, 2e-6)
    measured("partial_rows_sumsq_rel", e1, 2e-6)
    assert e0 <= 2e-6 and e1 <= 2e-6
    lib, st = _lib.lib(), _lib.current_stream()
    gamma = (0.75 + 0.5 * torch.rand(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    res = torch.randn(rows, C, generator=g).to(adt).to(DEV)
    for relu, use_res in ((1, False), (1, True), (0, False)):
        rd = res if use_res else None
        m_a, i_a = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        rm_a, rv_a = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y_a = torch.empty((rows, C), dtype=adt, device=DEV)
        want_mask = bool(bf16) and bool(relu) and C % 8 == 0          # bf16 + ReLU: both passes also leave the ReLU bit mask (one byte per 8 channels)
        mk_a = torch.zeros(rows * C // 8, dtype=torch.uint8, device=DEV) if want_mask else None
        mk_b = torch.zeros(rows * C // 8, dtype=torch.uint8, device=DEV) if want_mask else None
        _lib.check(lib.sp_bn_train_stats_from_conv(P(part[0]), P(part[1]), prow, part.shape[2], rows, C, 1e-5, 0.1, P(m_a), P(i_a), P(rm_a), P(rv_a), st), "fold")
        _lib.check(lib.sp_bn_apply_nhwc(P(z), int(bf16), P(m_a), P(i_a), P(gamma), P(beta), P(rd), P(y_a), rows, C, relu, P(mk_a), st), "apply")
        m_b, i_b = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        rm_b, rv_b = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y_b = torch.empty((rows, C), dtype=adt, device=DEV)
        _lib.check(lib.sp_bn_fold_apply_nhwc(P(z), int(bf16), P(part[0]), P(part[1]), prow, part.shape[2], rows, 1e-5, 0.1, P(gamma), P(beta), P(rd),
                                             P(y_b), rows, C, relu, P(m_b), P(i_b), P(rm_b), P(rv_b), P(mk_b), st), "fold+apply")
        torch.cuda.synchronize()
        assert torch.equal(m_a, m_b) and torch.equal(i_a, i_b) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
        assert torch.equal(y_a, y_b)
        if want_mask:
            bits = (y_b.float().reshape(-1, 8) > 0).to(torch.uint8)
            ref_mask = (bits << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(1).to(torch.uint8)
            assert torch.equal(mk_a, ref_mask) and torch.equal(mk_b, ref_mask)
        m64, v64 = z64.mean(0), z64.var(0, unbiased=False)
        em, ei = _rel(m_b, m64), _rel(i_b, 1 / torch.sqrt(v64 + 1e-5))
        measured("mean_rel", em, 1e-6)
        measured("invstd_rel", ei, 1e-6)
        assert em <= 1e-6 and ei <= 1e-6
        assert _rel(rm_b, 0.1 * m64) <= 2e-6 and _rel(rv_b, 0.9 + 0.1 * z64.var(0, unbiased=True)) <= 2e-6
        pre = (z64 - m64) / torch.sqrt(v64 + 1e-5) * gamma.double().cpu() + beta.double().cpu()
        if use_res:
            pre = pre + res.double().cpu()
        ref = torch.relu(pre) if relu else pre
        e = (_rel_bf16 if bf16 else _rel)(y_b.float(), ref)
        measured("y_rel", e, 4e-3 if bf16 else 3e-6)
        assert e <= (4e-3 if bf16 else 3e-6)


@pytest.mark.parametrize("bf16", [0, 1, 3, 7], ids=["fp32", "bf16", "bf16_grads", "bf16_grads_relu_mask"])
@pytest.mark.parametrize("two", [False, True], ids=["one_bn", "with_shortcut_bn"])
@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_dgrad_epilogue_sums_and_fused_fold_backward(case, two, bf16, measured):
    """dy = dgrad(dz_next) of a conv whose INPUT came out of BatchNorm + ReLU: the BSTATS epilogue's partial rows, folded by the stand-alone
    launch and by the fused pass, give the same (d gamma, d beta, dz, dres) bit for bit, and match float64 on the stored dy.
    "bf16_grads" (PoseTrainer grad_dtype "bf16"): the dgrad launch stores dy in bf16, its sums are those of the ROUNDED dy, dres is bf16."""
    from simple_pose_amd.train import Act
    name, I, O, k, s, p, H, W, B = case
    flag, g16, masked = bf16, bool(bf16 & 2), bool(bf16 & 4)     # masked: the ReLU source is the bit mask the forward pass leaves, not y
    bf16 = bool(bf16 & 1)
    if g16 and I % 8:
        pytest.skip("a bf16 NHWC gradient store needs c % 8 == 0")
    g = torch.Generator().manual_seed(9 + I + O)
    w = (torch.randn(O, I, k, k, generator=g, dtype=torch.float64) / np.sqrt(I * k * k)).float()
    adt = torch.bfloat16 if bf16 else torch.float32
    one = _OneLayer("conv", w, H, W, bf16, g16=g16, stride=s, pad=p)
    L = one.layer
    if not L.dgrad_full_cover and two:
        pytest.skip("stride-2 1x1: not a full-cover family")
    rows, C = B * H * W, I
    # the BatchNorm layer in front of this conv: saved z, its statistics, its output y (ReLU mask)
    zb = (torch.randn(rows, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + torch.randn(C, generator=g)).to(adt)
    z64 = zb.double()
    mean, invstd = z64.mean(0).float().to(DEV), (1 / torch.sqrt(z64.var(0, unbiased=False) + 1e-5)).float().to(DEV)
    gamma = (0.75 + 0.5 * torch.rand(C, generator=g)).to(DEV)
    yb = torch.relu((z64 - z64.mean(0)) * invstd.double().cpu() * gamma.double().cpu() + 0.1 * torch.randn(C, generator=g).double()).to(adt)
    z2 = (torch.randn(rows, C, generator=g) * 0.7 + 0.2).to(adt)
    mean2, invstd2 = z2.double().mean(0).float().to(DEV), (1 / torch.sqrt(z2.double().var(0, unbiased=False) + 1e-5)).float().to(DEV)
    zd, yd, z2d = zb.to(DEV).reshape(B, H, W, C), yb.to(DEV).reshape(B, H, W, C), z2.to(DEV).reshape(B, H, W, C)
    oh, ow = L.oh, L.ow
    dzn = torch.randn(B, oh, ow, L.c_out_buf, generator=g).to(adt).to(DEV)
    src = Act(yd, H, W, C)
    src.bn = (zd, mean, invstd)
    rsrc = yd
    if masked:
        bits = (yd.float().reshape(-1, 8) > 0).to(torch.uint8)
        src.mask = rsrc = (bits << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(1).to(torch.uint8).contiguous()
    if two:
        src.bn2 = (z2d, mean2, invstd2, "shortcut")
    dy = L.dgrad(dzn, B, None, bn_src=src)
    part, prow = src.bstats
    torch.cuda.synchronize()
    assert dy.dtype == (torch.bfloat16 if g16 else torch.float32)
    dy64 = dy.double().cpu().reshape(rows, C)
    g64 = dy64 * (yb.double() > 0)
    xh = (z64 - mean.double().cpu()) * invstd.double().cpu()
    sg, sgx = g64.sum(0), (g64 * xh).sum(0)
    lib, st = _lib.lib(), _lib.current_stream()
    # stand-alone fold + apply
    dga, dba = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.check(lib.sp_bn_bwd_sums_from_conv(P(part[0]), P(part[1]), prow, part.shape[2], C, P(dga), P(dba), st), "fold")
    dg2a = db2a = None
    if two:
        dg2a, db2a = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        _lib.check(lib.sp_bn_bwd_sums_from_conv(P(part[0]), P(part[2]), prow, part.shape[2], C, P(dg2a), P(db2a), st), "fold2")
    if two:                                # both folds in one launch (sp_bn_bwd_sums_from_conv_pair): the bits of the two calls
        pg, pb, pg2, pb2 = (torch.empty(C, device=DEV) for _ in range(4))
        _lib.check(lib.sp_bn_bwd_sums_from_conv_pair(P(part[0]), P(part[1]), P(part[2]), prow, part.shape[2], C, P(pg), P(pb), P(pg2), P(pb2), st), "pair")
        torch.cuda.synchronize()
        assert torch.equal(pg, dga) and torch.equal(pb, dba) and torch.equal(pg2, dg2a) and torch.equal(pb2, db2a)
    base = torch.randn(rows, C, generator=torch.Generator().manual_seed(3)).to(DEV).to(dy.dtype)
    dz_a, dres_a = torch.empty((rows, C), dtype=adt, device=DEV), base.clone()
    _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(dy), flag, P(rsrc), P(zd), P(mean), P(invstd), P(gamma), P(dga), P(dba), rows, rows, C, P(dz_a),
                                              P(dres_a), 1, st), "apply")
    # fused
    dgb, dbb = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dg2b = torch.empty(C, device=DEV) if two else None
    db2b = torch.empty(C, device=DEV) if two else None
    dz_b, dres_b = torch.empty((rows, C), dtype=adt, device=DEV), base.clone()
    _lib.check(lib.sp_bn_fold_bwd_apply_nhwc(P(dy), flag, P(rsrc), P(zd), P(part[0]), P(part[1]), P(part[2]) if two else None, prow, part.shape[2],
                                             P(mean), P(invstd), P(gamma), rows, rows, C, P(dgb), P(dbb), P(dg2b), P(db2b), P(dz_b), P(dres_b), 1, st),
               "fold+bwd apply")
    torch.cuda.synchronize()
    assert torch.equal(dga, dgb) and torch.equal(dba, dbb) and torch.equal(dz_a, dz_b) and torch.equal(dres_a, dres_b)
    if two:
        assert torch.equal(dg2a, dg2b) and torch.equal(db2a, db2b)
        xh2 = (z2.double() - mean2.double().cpu()) * invstd2.double().cpu()
        e2 = _rel(dg2b, (g64 * xh2).sum(0))
        measured("dgamma2_rel", e2, 3e-6)
        assert e2 <= 3e-6 and _rel(db2b, sg) <= 3e-6
    eg, eb = _rel(dgb, sgx), _rel(dbb, sg)
    measured("dgamma_rel", eg, 3e-6)
    measured("dbeta_rel", eb, 3e-6)
    assert eg <= 3e-6 and eb <= 3e-6
    ref_dz = gamma.double().cpu() * invstd.double().cpu() * (g64 - sg / rows - xh * sgx / rows)
    e = (_rel_bf16 if bf16 else _rel)(dz_b.float(), ref_dz)
    measured("dz_rel", e, 4e-3 if bf16 else 4e-6)
    assert e <= (4e-3 if bf16 else 4e-6)
    if g16:
        assert _rel_bf16(dres_b.float(), base.double().cpu() + g64) <= 4e-3
    else:
        assert _rel(dres_b - base, g64) <= 2e-6


# ---- the stem in training: bn1 + ReLU + maxpool as one forward pass / one backward entry ---------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 3], ids=["fp32", "bf16", "bf16_grads"])
def test_stem_bn_relu_maxpool_fused_matches_the_separate_passes(mode, measured):
    """sp_bn_apply_maxpool_nhwc == sp_bn_apply_nhwc + sp_maxpool3x3s2_idx_nhwc bit for bit (pooled map and winning taps), without the
    full-resolution activation; sp_bn_maxpool_bwd_nhwc == sp_maxpool3x3s2_bwd_idx_nhwc + sp_bn_train_bwd_nhwc (sums over the pooled grid
    instead of the full one: fp64 partials, so d gamma / d beta agree to fp32 rounding and dz with them)."""
    bf16, g16 = bool(mode & 1), bool(mode & 2)
    B, H, W, C = 3, 24, 20, 64
    g = torch.Generator().manual_seed(77)
    adt, gdt = (torch.bfloat16 if bf16 else torch.float32), (torch.bfloat16 if g16 else torch.float32)
    z = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.2).to(adt).to(DEV)
    rows = B * H * W
    z64 = z.double().reshape(rows, C)
    mean, invstd = z64.mean(0).float().contiguous(), (1 / torch.sqrt(z64.var(0, unbiased=False) + 1e-5)).float().contiguous()
    gamma, beta = (0.75 + 0.5 * torch.rand(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
    lib, st = _lib.lib(), _lib.current_stream()
    Ho, Wo = H // 2, W // 2
    # separate passes
    y = torch.empty_like(z)
    _lib.check(lib.sp_bn_apply_nhwc(P(z), mode & 1, P(mean), P(invstd), P(gamma), P(beta), None, P(y), rows, C, 1, None, st), "apply")
    pa, ia = torch.empty(B, Ho, Wo, C, dtype=adt, device=DEV), torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=DEV)
    _lib.check(lib.sp_maxpool3x3s2_idx_nhwc(P(y), mode & 1, P(pa), P(ia), B, H, W, C, st), "pool")
    # fused
    pb, ib = torch.empty_like(pa), torch.empty_like(ia)
    _lib.check(lib.sp_bn_apply_maxpool_nhwc(P(z), mode & 1, P(mean), P(invstd), P(gamma), P(beta), P(pb), P(ib), B, H, W, C, st), "apply+pool")
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ia, ib)
    # backward
    dyp = torch.randn(B, Ho, Wo, C, generator=g).to(gdt).to(DEV)
    ws = torch.empty(4 << 20, dtype=torch.uint8, device=DEV)
    dy = torch.empty(B, H, W, C, dtype=gdt, device=DEV)
    _lib.check(lib.sp_maxpool3x3s2_bwd_idx_nhwc(P(ia), P(dyp), mode, P(dy), B, H, W, C, st), "pool bwd")
    dz_a, dg_a, db_a = torch.empty_like(z), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.check(lib.sp_bn_train_bwd_nhwc(P(dy), mode, P(y), P(z), P(mean), P(invstd), P(gamma), rows, C, P(dz_a), P(dg_a), P(db_a), None, 0, P(ws), st), "bn bwd")
    dz_b, dg_b, db_b = torch.empty_like(z), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.check(lib.sp_bn_maxpool_bwd_nhwc(P(dyp), mode, P(ib), P(z), P(mean), P(invstd), P(gamma), P(beta), B, H, W, C, P(dg_b), P(db_b), P(dz_b), P(ws), st),
               "fused bwd")
    torch.cuda.synchronize()
    # (bf16 gradients: the separate path rounds the pooling's input gradient - a sum of up to four windows - to bf16 before the BatchNorm
    # reduction reads it; the fused path never forms that tensor, so the two differ by that rounding: the bar is bf16's, and the fused sums
    # are the more exact ones - checked against float64 below)
    sbar = 1e-2 if g16 else 2e-6
    eg, eb = _rel(dg_b, dg_a.double()), _rel(db_b, db_a.double())
    measured("stem_dgamma_rel", eg, sbar)
    assert eg <= sbar and eb <= sbar, (eg, eb)
    e = (_rel_bf16 if bf16 else _rel)(dz_b.float(), dz_a.double())
    zbar = 1.2e-2 if g16 else (4e-3 if bf16 else 4e-6)
    measured("stem_dz_rel", e, zbar)
    assert e <= zbar
    # float64 reference of the sums from the pooled gradient itself
    yy = y.double().cpu().reshape(B, H, W, C)
    gfull = torch.zeros(B, H, W, C, dtype=torch.float64)
    iac, dypc = ia.cpu().long(), dyp.double().cpu()
    for oy in range(Ho):
        for ox in range(Wo):
            t = iac[:, oy, ox, :]
            iy, ix = 2 * oy - 1 + t // 3, 2 * ox - 1 + t % 3
            bi = torch.arange(B)[:, None].expand(B, C)
            ci = torch.arange(C)[None, :].expand(B, C)
            gfull.index_put_((bi, iy, ix, ci), dypc[:, oy, ox, :], accumulate=True)
    gm = (gfull * (yy > 0)).reshape(rows, C)
    xh = (z64.cpu() - mean.double().cpu()) * invstd.double().cpu()
    assert _rel(db_b, gm.sum(0)) <= 3e-6 and _rel(dg_b, (gm * xh).sum(0)) <= 3e-6


def test_pixel_unshuffle_bf16_is_the_fp32_permutation():
    """sp_pixel_unshuffle2_nhwc_bf16 (bf16 activation gradients through the DUC head's PixelShuffle) moves the same elements as the fp32 kernel."""
    B, h, w, C = 3, 6, 5, 64
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, 2 * h, 2 * w, C // 4, generator=g).to(torch.bfloat16).to(DEV)
    lib, st = _lib.lib(), _lib.current_stream()
    a = torch.empty(B, h, w, C, device=DEV)
    _lib.check(lib.sp_pixel_unshuffle2_nhwc(P(dy.float().contiguous()), P(a), B, h, w, C, st), "fp32")
    b = torch.empty(B, h, w, C, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.sp_pixel_unshuffle2_nhwc_bf16(P(dy), P(b), B, h, w, C, st), "bf16")
    torch.cuda.synchronize()
    assert torch.equal(a, b.float())
    ref = torch.nn.functional.pixel_unshuffle(dy.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(a, ref.contiguous())


# ---------------------------------------------------------------------------------------------- grouped convolutions (resnext*, round 6)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C,groups,stride,B,H,W", [(128, 32, 1, 3, 12, 10), (256, 32, 2, 2, 16, 12), (512, 32, 1, 2, 9, 7), (1024, 32, 2, 2, 8, 6),
                                                   (2048, 32, 1, 1, 5, 4)])
def test_grouped_conv_forward_dgrad_wgrad_against_float64(C, groups, stride, B, H, W, dtype):
    """nn.Conv2d(C, C, 3, stride, padding=1, groups=32) (nets/pose_resnet_dconv.py:97-101) through ConvT's grouped lowering, one layer alone:
    forward (sp_conv2d_fwd with sp_conv_desc.c_in_group on panels from sp_pack_conv_weights_grouped_taps), input gradient (the same launch on
    transposed / flipped panels; stride 2: one launch per output phase) and weight gradient (sp_conv2d_wgrad_grouped: partial sums per row chunk,
    fixed-order fp64 fold) against torch's float64 grouped convolution on the same (bf16-rounded) operands; group widths 4 ... 64."""
    import types
    from simple_pose_amd.train import ConvT, FlatParams
    bf = dtype == "bf16"
    adt = torch.bfloat16 if bf else torch.float32
    cpg = C // groups
    conv = torch.nn.Conv2d(C, C, 3, stride=stride, padding=1, groups=groups, bias=False)
    wn = synth.tensor_normal(3, f"gconv{C}/w", (C, cpg, 3, 3), std=(2.0 / (cpg * 9)) ** 0.5)
    conv.weight.data.copy_(torch.from_numpy(wn).to(adt).float())
    holder = torch.nn.Module()
    holder.add_module("g", conv)
    holder = holder.to(DEV)
    flat = FlatParams(holder)
    tr = types.SimpleNamespace(bf16=bf, g16=bf, grad_dtype=adt, flat=flat, kernel_events=None)
    layer = ConvT(tr, "g", "conv", holder.g.weight.detach(), H, W, stride=stride, pad=1, groups=groups)
    assert layer.d_fwd.c_in_group == 64 and len(layer.d_dgrad) == (1 if stride == 1 else 4)
    layer.pack_grouped(_lib.current_stream())
    x = torch.from_numpy(synth.tensor_normal(3, f"gconv{C}/x", (B, C, H, W))).to(adt)
    OH, OW = layer.oh, layer.ow
    dz = torch.from_numpy(synth.tensor_normal(3, f"gconv{C}/dz", (B, C, OH, OW))).to(adt)
    xd = x.double().requires_grad_(True)
    wd = torch.from_numpy(wn).to(adt).double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, wd, stride=stride, padding=1, groups=groups)
    ref.backward(dz.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dzg = dz.permute(0, 2, 3, 1).contiguous().to(DEV)
    z = layer.forward(xg, B)
    dx = layer.dgrad(dzg, B, None)
    layer.wgrad_grouped(xg, dzg, B)
    torch.cuda.synchronize()
    tol = 2e-2 if bf else 2e-5                       # (bf16: outputs are rounded to bf16; the weight gradient is fp32 either way)
    got = z.float().cpu().permute(0, 3, 1, 2).double()
    assert float((got - ref.detach()).abs().max() / ref.detach().abs().max()) < tol
    gdx = dx.float().cpu().permute(0, 3, 1, 2).double()
    assert float((gdx - xd.grad).abs().max() / xd.grad.abs().max()) < tol
    gdw = flat.view("g.weight", grad=True).cpu().double().view(C, cpg, 3, 3)
    assert float((gdw - wd.grad).abs().max() / wd.grad.abs().max()) < 2e-5
    # accumulate form of the input gradient (a second consumer's share already in place), and bit-reproducibility of the weight gradient
    acc0 = torch.from_numpy(synth.tensor_normal(3, f"gconv{C}/acc", (B, H, W, C))).to(adt).to(DEV)
    dx2 = layer.dgrad(dzg, B, acc0.clone())
    torch.cuda.synchronize()
    want = (acc0.float().cpu().double() + xd.grad.permute(0, 2, 3, 1)).abs().max()
    assert float((dx2.float().cpu().double() - (acc0.float().cpu().double() + xd.grad.permute(0, 2, 3, 1))).abs().max() / want) < tol
    first = flat.view("g.weight", grad=True).clone()
    layer.wgrad_grouped(xg, dzg, B)
    torch.cuda.synchronize()
    assert torch.equal(first, flat.view("g.weight", grad=True))
