"""CPU tests of the host side: state_dict layout, weight packing + launch descriptors (through the CPU descriptor
interpreter in tests/desc_interp.py) against the oracle forward, and the C-ABI surface of the built library."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import nets_oracle
from simple_pose_amd import _lib, engine, synth
from simple_pose_amd.build import LIB_PATH
from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc
from tests.desc_interp import TorchPacker, run_program_cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mod,head,nkeys,nparams", [(pose_resnet_dconv, "dconv", 338, 33999697),
                                                     (pose_resnet_duc, "duc", 332, 29428945)])
def test_state_dict_layout_matches_reference(mod, head, nkeys, nparams):
    m = mod.resnet50(pretrained=False, num_classes=17)
    mine = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
    assert len(mine) == nkeys
    assert set(mine) == set(nets_oracle.state_dict_shapes_resnet50(head))
    assert sum(p.numel() for p in m.parameters()) == nparams
    # checkpoint convention of the reference: {"ema": state_dict, "epoch": n}, keys optionally prefixed "module."
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head)).items()}
    m.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("head", ["dconv", "duc"])
def test_packing_and_descriptors_reproduce_oracle_forward(head):
    """Program (packed weights + descriptors) interpreted on CPU == oracle forward, on a 64x64 crop, B=2."""
    shapes = nets_oracle.state_dict_shapes_resnet50(head)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=3).items()}
    x = torch.from_numpy(synth.input_images(2, seed=3, h=64, w=64))
    prog = engine.resnet_program(sd, head, in_h=64, in_w=64, packer=TorchPacker())
    with torch.no_grad():
        ref = nets_oracle.FORWARDS["resnet50_" + head](sd, x)
        got, _ = run_program_cpu(prog, x)
    assert got.shape == ref.shape == (2, 17, 16, 16)
    rel = (got - ref).abs().max() / ref.abs().max()
    assert rel < 1e-5, rel
    # algorithmic FLOPs bookkeeping: per-image conv MACs scale with the image area
    full = engine.resnet_program(sd, head, in_h=256, in_w=192, packer=TorchPacker())
    expect = {"dconv": 10.8528e9, "duc": 11.7517e9}[head]
    assert abs(full.flops_per_image - expect) / expect < 1e-4, full.flops_per_image


VARIANTS = [("resnet18", "dconv", False), ("resnet34", "duc", False), ("wide_resnet50_2", "dconv", False), ("resnet18", "dconv", True),
            ("resnext50_32x4d", "dconv", False), ("resnext101_32x8d", "duc", False)]      # (round 5: the grouped resnext factories, golden g12)


def _variant_model(arch, head, se):
    return getattr(pose_resnet_dconv if head == "dconv" else pose_resnet_duc, arch)(pretrained=False, num_classes=17, reduction=se)


@pytest.mark.parametrize("arch,head,se", VARIANTS, ids=[f"{a}_{h}" + ("_se" if s else "") for a, h, s in VARIANTS])
def test_resnet_variant_layout_and_lowering(golden, arch, head, se):
    """The reference's other ResNet factories (nets/pose_resnet_dconv.py:282-403: BasicBlock nets resnet18 / 34, wide_resnet*_2): the module
    tree has the reference's state_dict keys and shapes in the reference's order (key lists frozen in g11 from the real reference), and the
    lowered program interpreted on the CPU equals the oracle forward."""
    g = golden("g12_resnext.npz" if arch.startswith("resnext") else "g11_resnet_variants.npz")
    tag = f"{arch}_{head}" + ("_se" if se else "")
    m = _variant_model(arch, head, se)
    sd0 = m.state_dict()
    assert list(sd0.keys()) == list(g[f"{tag}/keys"])
    assert [",".join(str(d) for d in v.shape) for v in sd0.values()] == list(g[f"{tag}/shapes"])
    layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in sd0.items()]
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(layout, seed=3).items()}
    x = torch.from_numpy(synth.input_images(2, seed=3, h=64, w=64))
    prog = engine.resnet_program(sd, head, in_h=64, in_w=64, blocks=m.BLOCKS, packer=TorchPacker())
    with torch.no_grad():
        ref = nets_oracle.FORWARDS["resnet50_" + head](sd, x)
        got, _ = run_program_cpu(prog, x)
    assert got.shape == ref.shape == (2, 17, 16, 16)
    rel = (got - ref).abs().max() / ref.abs().max()
    assert rel < 1e-5, rel
    if arch.startswith("resnext"):          # grouped 3x3 (groups = 32): block-diagonal panels, K per tap = the panel (sp_conv_desc.c_in_group)
        grouped = [op for op in prog.ops if op.kind == "conv" and op.desc.c_in_group]
        assert len(grouped) == sum(m.BLOCKS) and all(op.desc.tile_n == op.desc.c_in_group and op.desc.k_pad == 9 * op.desc.c_in_group for op in grouped)


def test_grouped_nets_are_accepted_for_training_and_refuse_the_cpu():
    """resnext*: the eval forward is lowered since round 5 (grouped implicit GEMM), the train step since round 6 (grouped dgrad launches on
    transposed panels + sp_conv2d_wgrad_grouped: tests/test_gpu_train.py::test_grouped_nets_train_step_vs_oracle).  Without a GPU the trainer
    still refuses - for the reason every net is refused on the CPU (no fallback path), not because the net is grouped."""
    from simple_pose_amd.train import PoseTrainer
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu tests on a GPU box")
    with pytest.raises(Exception) as e:
        PoseTrainer(pose_resnet_dconv.resnext50_32x4d(num_classes=17))
    assert "grouped" not in str(e.value)


def test_buffer_plan_never_aliases_live_tensors():
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv")
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=1).items()}
    prog = engine.resnet_program(sd, "dconv", in_h=64, in_w=64, packer=TorchPacker())
    bufs = prog._alloc(1, torch.device("cpu"))
    live = {}
    last = {}
    for i, op in enumerate(prog.ops):
        for nm in (op.src, op.res, op.dst):
            if nm:
                last[nm] = i
    for i, op in enumerate(prog.ops):
        if op.dst in bufs:
            p = bufs[op.dst].data_ptr()
            for nm, (q, until) in live.items():
                assert not (q == p and until >= i and nm != op.dst), (op.name, nm)
            live[op.dst] = (p, last[op.dst])
    total = sum(t.numel() for t in {b.data_ptr(): b for b in bufs.values()}.values()) * 4
    naive = sum(np.prod(prog.shapes[o.dst]) for o in prog.ops if o.dst != "heat") * 4
    assert total < 0.35 * naive  # exact-size reuse only; ~27 % of the no-reuse footprint


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a CPU-only box and exports exactly what include/simple_pose_hip.h declares."""
    assert os.path.isfile(LIB_PATH), "build first: python -m simple_pose_amd.build"
    hdr = open(os.path.join(ROOT, "include", "simple_pose_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(sp_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    handle = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.sp_abi_version() == _lib.ABI_VERSION
    assert ctypes.sizeof(_lib.ConvDesc) == 31 * 4          # 25 geometry ints, flags, tile_m, tile_n, stride_x, kernel, c_in_group


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    lib = _lib.lib()
    d = _lib.ConvDesc()
    rc = lib.sp_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.sp_last_error()
    one = ctypes.c_void_p(16)
    d.batch = d.in_h = d.in_w = d.grid_h = d.grid_w = d.c_out = 1
    d.c_in = 3
    rc = lib.sp_conv2d_fwd(ctypes.byref(d), one, one, None, None, None, one, None)
    assert rc == -1 and b"multiple of 4" in lib.sp_last_error()
    rc = lib.sp_decode_gauss_taylor(one, one, 1, 17, 64, 48, 12, one, one, None)
    assert rc == -1 and b"odd" in lib.sp_last_error()


def test_product_path_refuses_cpu_tensors():
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17).eval()
    with pytest.raises(_lib.HipLibraryError):
        m(torch.zeros(1, 3, 256, 192))
    from simple_pose_amd.metrics import GaussTaylorKeyPointDecoder
    with pytest.raises(_lib.HipLibraryError):
        GaussTaylorKeyPointDecoder()(torch.zeros(1, 17, 64, 48), torch.zeros(1, 2, 3))


def test_hrnet_module_layout_and_lowering_reproduce_oracle(golden):
    """HRNet-W32: product module has the reference's 1,754 keys in order; its Program, interpreted on CPU, equals the
    oracle forward on a 64x64 crop."""
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    g = golden("g3_hrnet_w32_fwd.npz")
    m = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sd0 = m.state_dict()
    assert list(sd0.keys()) == list(g["keys"]) and len(sd0) == 1754
    assert sum(p.numel() for p in m.parameters()) == 28536113 and len(list(m.parameters())) == 878
    shapes = hrnet_state_dict_shapes(m.cfg, 17)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=2).items()}
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth.input_images(1, seed=2, h=64, w=64))
    prog = engine.hrnet_program(sd, m.cfg, in_h=64, in_w=64, packer=TorchPacker())
    with torch.no_grad():
        ref = nets_oracle.hrnet_forward(sd, x, m.cfg)
        got, _ = run_program_cpu(prog, x)
    assert got.shape == ref.shape == (1, 17, 16, 16)
    assert (got - ref).abs().max() / ref.abs().max() < 1e-5
    full = engine.hrnet_program(sd, m.cfg, in_h=256, in_w=192, packer=TorchPacker())
    assert abs(full.flops_per_image - 15.29e9) / 15.29e9 < 1e-3     # 15.2900 GFLOP / image
    with pytest.raises(_lib.HipLibraryError):
        m.eval()(torch.zeros(1, 3, 256, 192))


def test_se_variant_module_layout_and_lowering(golden):
    """reduction=True: reference key order (SE after downsample), Program interpreted on CPU == oracle."""
    g = golden("g1s_dconv_se_fwd.npz")
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
    assert list(m.state_dict().keys()) == list(g["keys"]) and sum(p.numel() for p in m.parameters()) == 45148497
    shapes = nets_oracle.state_dict_shapes_resnet50("dconv", se=True)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=4).items()}
    x = torch.from_numpy(synth.input_images(2, seed=4, h=64, w=64))
    prog = engine.resnet_program(sd, "dconv", in_h=64, in_w=64, packer=TorchPacker())
    with torch.no_grad():
        ref = nets_oracle.resnet_dconv_forward(sd, x)
        got, _ = run_program_cpu(prog, x)
    assert (got - ref).abs().max() / ref.abs().max() < 1e-5


@pytest.mark.parametrize("head", ["dconv", "duc"])
def test_bf16_lowering_tracks_fp32_oracle(head):
    """bf16 program (bf16 operands, fp32 accumulate) interpreted on CPU stays within bf16 tolerance of the fp32 oracle."""
    shapes = nets_oracle.state_dict_shapes_resnet50(head)
    sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(shapes, seed=3).items()}
    x = torch.from_numpy(synth.input_images(1, seed=3, h=64, w=64))
    prog = engine.resnet_program(sd, head, in_h=64, in_w=64, dtype="bf16", packer=TorchPacker())
    assert all(op.w.dtype == torch.bfloat16 for op in prog.ops if op.kind == "conv")
    with torch.no_grad():
        ref = nets_oracle.FORWARDS["resnet50_" + head](sd, x)
        got, _ = run_program_cpu(prog, x)
    assert got.dtype == torch.float32
    rel = (got - ref).abs().max() / ref.abs().max()
    assert 1e-4 < rel < 5e-2, rel     # CPU bf16 autocast vs fp32 = 1.07e-2


def test_gradient_buckets_tile_the_flat_buffer_in_reverse_parameter_order():
    """DDP-style reducer plan (simple_pose_amd.train.PoseTrainer._plan_buckets): contiguous slices, cut at parameter
    boundaries from the end of the buffer, every parameter in exactly one bucket; a bucket fires when its last gradient lands."""
    import torch
    from simple_pose_amd.train import FlatParams, PoseTrainer

    m = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 3), torch.nn.BatchNorm2d(64), torch.nn.Conv2d(64, 64, 3), torch.nn.Conv2d(64, 17, 1))
    shim = PoseTrainer.__new__(PoseTrainer)
    shim.flat = FlatParams(m)
    shim._plan_buckets(bucket_mb=4096 * 4 / (1 << 20))            # 4096-float buckets
    names = [n for n, _ in m.named_parameters()]
    assert shim.buckets[0]["hi"] == shim.flat.numel and shim.buckets[-1]["lo"] == 0
    for a, b in zip(shim.buckets, shim.buckets[1:]):
        assert a["lo"] == b["hi"]
    assert sorted(n for b in shim.buckets for n in b["names"]) == sorted(names)
    assert "3.bias" in shim.buckets[0]["names"] and "0.weight" in shim.buckets[-1]["names"]
    # the net's first three parameters (the stem: conv weight + BatchNorm pair, whose gradients exist last) have a bucket of their own
    assert shim.buckets[-1]["names"] == {"0.weight", "0.bias", "1.weight"} and shim.buckets[-1]["lo"] == 0
    for b in shim.buckets:
        for n in b["names"]:
            o, k = shim.flat.offsets[n]
            assert b["lo"] <= o and o + k <= b["hi"]
    assert len(shim.buckets) >= 2
    shim.world = 1
    shim._grads_ready("3.bias")                                    # single rank: nothing to launch, no state needed


def test_solver_schedule_and_config_schema_match_the_reference():
    """simple_pose_amd.processors.ddp_pose_resnet_solver: MultiStepLR in closed form == torch's scheduler (ddp...:73-77), and the
    shipped yaml carries every key the reference's solver reads (configs/ddp_fast_pose.yaml)."""
    import os
    import torch
    import yaml
    from simple_pose_amd.processors.ddp_pose_resnet_solver import AverageLogger, multi_step_lr

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.1)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3, 5], gamma=0.1)
    for epoch in range(8):
        assert abs(opt.param_groups[0]["lr"] - multi_step_lr(0.1, [3, 5], 0.1, epoch)) < 1e-12
        opt.step(); sch.step()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "simple_pose_amd", "configs", "ddp_fast_pose.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    assert {"model_name", "data", "model", "optim", "val", "gpus"} <= set(cfg)
    assert {"batch_size", "num_workers", "debug"} <= set(cfg["data"])
    assert {"type", "name", "num_joints", "pretrained"} <= set(cfg["model"])
    assert {"lr", "amp", "sync_bn", "milestones", "epochs", "gamma"} <= set(cfg["optim"])
    assert {"interval", "weight_path"} <= set(cfg["val"])
    lg = AverageLogger()
    for v in (1.0, 2.0, 6.0):
        lg.update(torch.tensor(v))
    assert lg.avg() == 3.0


def test_bench_self_launch_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` is its own launcher (no torch.distributed.run needed): the parent starts 2 fresh ranks before it
    imports torch.  On a machine without a GPU every rank exits with the "no CPU path" message and the launcher returns non-zero
    instead of hanging or printing a line."""
    import subprocess
    import sys

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu test on a GPU box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    # at least one rank refused; normally both do (the launcher gives the survivors a few seconds to say so before it terminates them),
    # but under load the second interpreter may still be importing when the grace period ends - that is not what this test is about
    assert r.stderr.count("no CPU path") >= 1


def test_bench_micro_mode_hands_its_summary_to_emit():
    """`bench.py --mode micro` calls tools/bench_micro.main(argv, emit=bench.emit): after `claim_stdout()` file descriptor 1 is stderr, so a
    bare print() of the summary would leave stdout empty (round-4 advisor finding).  Without a GPU: the plumbing, with a stand-in for the
    measurement module (the real run is `tests/test_gpu_parity.py::test_bench_micro_mode_prints_one_json_line_on_stdout`)."""
    import inspect
    import json
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import bench_micro
    finally:
        sys.path.pop(0)
    assert "emit" in inspect.signature(bench_micro.main).parameters
    code = (
        "import sys, types\n"
        "sys.argv = ['bench.py', '--mode', 'micro']\n"
        "m = types.ModuleType('bench_micro')\n"
        "def main(argv, emit=None):\n"
        "    print('noise from a library on fd 1')\n"
        "    emit({'metric': 'stand-in', 'rows': 3})\n"
        "    return 0\n"
        "m.main = main\n"
        "sys.modules['bench_micro'] = m\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "bench.main()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip()] == [{"metric": "stand-in", "rows": 3}]
    assert "noise from a library" in r.stderr


def test_bench_dry_launch_eight_ranks_over_gloo():
    """`python bench.py --gpus 8 --dry-launch`: the launcher starts 8 fresh interpreters with the env:// contract of a real run; they
    rendezvous over gloo, agree on the rank table, run the measurement's SUM / MAX reductions on known numbers and the sampler rule,
    rank 0 prints ONE JSON line.  A rank that dies after the rendezvous ends the job with its exit code instead of a hang."""
    import json
    import subprocess
    import sys

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SP_BENCH_DRY_FAIL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch", "--batch", "32", "--steps", "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_launch"] and d["n_ranks"] == 8 and d["ranks_seen"] == list(range(8)) and d["reductions_ok"]
    assert d["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                 # the explicit default, recorded
    # the world-8 train step's exchange plan, built on every rank without a GPU: 4 gradient buckets of >= 32 MiB cut at parameter
    # boundaries (40.0 / 34.0 / 32.6 / 23.0 MiB = the 136 MB of fp32 gradients, ddp...:91-93) + since round 5 the stem's own bucket (conv1.weight,
    # bn1.weight, bn1.bias: 0.04 MiB - its gradients exist last, so layer1's bucket no longer waits for them) and 52 + 52 SyncBatchNorm
    # messages (ddp...:89-90; what PoseTrainer.collective_count reaches on the GPU)
    plan = d["train_step_plan"]
    assert plan["gradient_buckets"] == 5 and plan["bucket_mbytes"][-1] < 0.1 and plan["sync_bn_messages_per_step"] == 104 and plan["batchnorm_layers"] == 56
    assert plan["same_on_every_rank"] and abs(sum(plan["bucket_mbytes"]) - 4 * plan["gradient_floats"] / (1 << 20)) < 0.5
    assert plan["communicators"] == 1
    n_dev = torch.cuda.device_count()
    assert d["one_device_per_rank"] == (n_dev >= 8)
    assert "other_configs" not in d                              # (not the driver's command: --batch 32; the three-job flow is the next test)
    r = subprocess.run(cmd + ["--hsa-ipc-legacy", "1"], env=dict(env, SP_BENCH_DRY_FAIL_RANK="5"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _dry_three_jobs(n, hook, torchrun=False, deadline="25"):
    import json
    import subprocess
    import sys
    import time

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SP_BENCH_DRY_FAIL_RANK", "SP_NATIVE_COMM", "SP_BENCH_CHILD"):
        env.pop(k, None)
    env.update(SP_BENCH_TEST_SELF_CHECK=hook, SP_BENCH_CHECK_DEADLINE_S=deadline)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-launch", "--steps", "3", "--warmup", "1"]
    if torchrun:
        from simple_pose_amd.launch import free_port
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]                 # the headline's exit code: whatever became of the extra jobs
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_launch"] and d["n_ranks"] == n and d["reductions_ok"]
    assert list(d)[-1] == "other_configs" and len(d["other_configs"]) == 1
    return d["other_configs"][0], wall


@pytest.mark.parametrize("n", [2, 8])
def test_bench_dry_launch_runs_the_three_jobs_with_deadlines(n):
    """The driver's command at N > 1 (`bench.py --gpus N`, nothing overridden), over gloo without a GPU: the supervisor - which touches neither
    torch nor a GPU - runs three N-rank jobs in order, each under a hard deadline: the inference replicas (the headline line), the collective
    self-check, and the bf16 train step (BASELINE config 4's batch-sharded step; reference: processors/ddp_pose_resnet_solver.py:36,85-93,110-133)
    with the self-check's verdict handed in, appended to the line as `other_configs`.  The self-check's local verdict is injected:
      pass  -> every rank agrees, the train job takes the native path;
      hang  -> rank 1 never returns (what code that has never met a peer does): the job costs its deadline, its ranks are killed, the train
               job runs over torch.distributed and the line says why - exit code 0, well inside the dry run's time limit;
      raise -> rank 1 cannot complete the comparison: it exits 13 without voting, the supervisor ends the others (round-5 advisor finding)."""
    oc, _ = _dry_three_jobs(n, "pass")
    assert oc["n_gpus"] == n and oc["job"]["status"] == "ok" and oc["global_batch"] == 32 * n and oc["dtype"] == "bf16"
    assert oc["collective_self_check"]["native"] and oc["collective_self_check"]["job"]["status"] == "ok"
    assert oc["native_flag_agreed"] is True and "sp_comm" in oc["collective_path"]
    dl = "10" if n == 2 else "15"
    oc, wall = _dry_three_jobs(n, "hang", deadline=dl)
    chk = oc["collective_self_check"]
    assert chk["job"]["status"] == "timeout" and "hung" in chk["reason"] and f"{dl} s deadline" in chk["reason"] and not chk["native"]
    assert oc["job"]["status"] == "ok" and oc["collective_path"] == "torch.distributed" and oc["native_flag_agreed"] is False
    assert wall < 150.0
    if n == 2:
        oc, _ = _dry_three_jobs(n, "raise")
        chk = oc["collective_self_check"]
        assert chk["job"]["status"] == "died" and "rank 1" in chk["reason"] and "13" in chk["reason"] and oc["collective_path"] == "torch.distributed"
        oc, _ = _dry_three_jobs(n, "fail")                     # a completed comparison that found a difference on one rank: agreed "no"
        assert oc["collective_self_check"]["job"]["status"] == "ok" and not oc["collective_self_check"]["native"]
        assert "bit for bit" in oc["collective_self_check"]["reason"] and oc["collective_path"] == "torch.distributed"


def test_bench_under_torchrun_each_worker_supervises_its_own_rank():
    """The driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: every worker then supervises the ONE child
    of its rank; the supervisors meet through a shared directory (exit codes, the agreed decision) and the extra jobs rendezvous in a
    FileStore there (torchrun's store belongs to the main job).  Same three jobs, same outcomes, one line from rank 0."""
    oc, _ = _dry_three_jobs(2, "pass", torchrun=True)
    assert oc["job"]["status"] == "ok" and oc["collective_self_check"]["native"] and "sp_comm" in oc["collective_path"]
    oc, _ = _dry_three_jobs(2, "raise", torchrun=True)
    chk = oc["collective_self_check"]
    assert chk["job"]["status"] == "died" and not chk["native"] and oc["collective_path"] == "torch.distributed" and oc["job"]["status"] == "ok"
    oc, wall = _dry_three_jobs(2, "hang", torchrun=True, deadline="10")
    assert oc["collective_self_check"]["job"]["status"] == "timeout" and oc["collective_path"] == "torch.distributed" and wall < 120.0


def test_conv_kernel_name_query_matches_dispatch():
    """sp_conv2d_kernel_name (used by bench.py's roofline object and the profile summaries) names what the dispatch launches."""
    d = _lib.ConvDesc()
    d.batch, d.in_h, d.in_w, d.c_in = 8, 16, 12, 256
    d.grid_h, d.grid_w, d.c_out, d.n_pad = 16, 12, 256, 256
    d.taps_h, d.taps_w, d.k_pad, d.stride = 3, 3, 2304, 1
    d.dy0, d.dy_step, d.dx0, d.dx_step = -1, 1, -1, 1
    d.out_h, d.out_w, d.out_c = 16, 12, 256
    d.oy_mul = d.ox_mul = 1
    d.phases_y = d.phases_x = 1
    d.flags = _lib.SP_CONV_BF16 | _lib.SP_CONV_RELU
    d.tile_m, d.tile_n, d.kernel = 128, 128, _lib.SP_CONV_KERNEL_IGEMM
    assert _lib.conv_kernel_name(d, True) == "conv_igemm_kernel<128, 128, 2, 2, true, true, true, false, true, false>"   # 36 K tiles: deep ring
    d.tile_m, d.tile_n, d.kernel = 128, 256, _lib.SP_CONV_KERNEL_RING
    assert _lib.conv_kernel_name(d, False) == "conv_ring_kernel<128, 256, 2, 4, 3, false, 0>"
    d.kernel = _lib.SP_CONV_KERNEL_RING_LW            # the same ring fed by four loader waves (round 5): last template argument
    assert _lib.lib().sp_conv2d_ring_ok(d) == 1
    assert _lib.conv_kernel_name(d, True) == "conv_ring_kernel<128, 256, 2, 4, 3, true, 4>"
    d.tile_m, d.tile_n = 256, 256                     # 221 VGPRs: no loader-wave instantiation (three waves per SIMD)
    assert _lib.lib().sp_conv2d_ring_ok(d) == 0
    d.kernel = _lib.SP_CONV_KERNEL_RING
    assert _lib.lib().sp_conv2d_ring_ok(d) == 1
    d.tile_m, d.tile_n, d.kernel = 192, 128, _lib.SP_CONV_KERNEL_RING_LW4    # four MFMA waves (2 x 2 of 96x64) + four loader waves (round 6)
    assert _lib.lib().sp_conv2d_ring_ok(d) == 1
    assert _lib.conv_kernel_name(d, False) == "conv_ring_kernel<192, 128, 2, 2, 3, false, 4>"
    d.tile_m = 96                                                           # a tile eight waves cannot cut: 1 x 4 waves of 96x32
    assert _lib.lib().sp_conv2d_ring_ok(d) == 1 and _lib.conv_kernel_name(d, True) == "conv_ring_kernel<96, 128, 1, 4, 4, true, 4>"
    d.kernel = _lib.SP_CONV_KERNEL_RING_LW
    assert _lib.lib().sp_conv2d_ring_ok(d) == 0                             # (no 96-row tile with eight MFMA waves)
    d.flags = _lib.SP_CONV_RELU
    d.tile_m, d.tile_n, d.kernel = 64, 128, _lib.SP_CONV_KERNEL_IGEMM
    d.c_in, d.k_pad = 256, 2304
    assert _lib.conv_kernel_name(d, False).startswith("conv_igemm_kernel<64, 128, 2, 2, true, false, false")
    # the fused blocks name the kernel their own dispatch selects (variant 4 = sp_basic_block_c32, 5 = sp_bottleneck_c64, 6 = sp_basic_block_c64; `d` = the block's 3x3 conv)
    for variant, c, k_pad, name in ((4, 32, 320, "basic_block_c32_w8_kernel"), (5, 64, 576, "bottleneck_c64_w8_kernel"), (6, 64, 576, "basic_block_c64_kernel")):
        d = _lib.ConvDesc()
        d.batch, d.in_h, d.in_w, d.c_in = 2, 64, 48, c
        d.grid_h, d.grid_w, d.c_out, d.n_pad = 64, 48, c, 64
        d.taps_h, d.taps_w, d.k_pad, d.stride = 3, 3, k_pad, 1
        d.dy0, d.dy_step, d.dx0, d.dx_step = -1, 1, -1, 1
        d.out_h, d.out_w, d.out_c = 64, 48, c
        d.oy_mul = d.ox_mul = 1
        d.phases_y = d.phases_x = 1
        d.flags = _lib.SP_CONV_BF16 | _lib.SP_CONV_RELU
        ok = {4: _lib.lib().sp_basic_block_c32_ok, 5: _lib.lib().sp_bottleneck_c64_ok, 6: _lib.lib().sp_basic_block_c64_ok}[variant](d)
        assert ok == 1 and _lib.conv_kernel_name(d, False, variant) == name


def test_hip_packer_has_no_host_path():
    """engine.HipPacker is the C ABI on device memory only: CPU tensors are refused loudly (the torch restatement of the layouts lives in
    tests/desc_interp.py and is never imported by the package)."""
    import simple_pose_amd
    w = torch.zeros((8, 4, 3, 3))
    with pytest.raises(_lib.HipLibraryError):
        engine.HipPacker().conv(w)
    with pytest.raises(_lib.HipLibraryError):
        engine.HipPacker().fold_bn(torch.ones(4), torch.zeros(4), torch.zeros(4), torch.ones(4))
    src = open(os.path.join(ROOT, "simple_pose_amd", "engine.py")).read()
    assert "desc_interp" not in src.replace("tests/desc_interp.TorchPacker restates", "") and "import oracle" not in src


def test_stream_pin_is_per_thread_and_per_device():
    """`_lib.pin_stream` (the train tape's shortcut around torch.cuda.current_stream) is visible only to the pinning thread and only for
    the pinned device (round-4 advisor finding: a process-global pin handed the trainer's stream to every caller)."""
    import threading

    prev = _lib.pin_stream((ctypes.c_void_p(0x1234), 0))
    try:
        assert _lib.current_stream().value == 0x1234                              # the pinning thread, "current device"
        assert _lib.current_stream(torch.device("cuda", 0)).value == 0x1234       # ... and the pinned device by name
        seen = {}

        def other():
            seen["pin"] = getattr(_lib._pin, "value", None)
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert seen["pin"] is None                                               # another thread asks torch, as before
        if not torch.cuda.is_available():
            with pytest.raises(Exception):
                _lib.current_stream(torch.device("cuda", 1))                    # another device: not the pin -> torch (no GPU here: raises)
    finally:
        _lib.pin_stream(prev)
    assert getattr(_lib._pin, "value", None) is prev
