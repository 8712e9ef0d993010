#!/usr/bin/env python3
"""bench.py - images/sec of forward + GaussTaylor decode, ResNet50-DConv 256x192, bs=128 per GPU, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128] [--arch dconv|duc] [--no-cpu-baseline]

One "step" = one pass of the hot path (network forward + key-point decode) over one batch of synthetic images that
are already resident in HBM.  N > 1: one process per GPU - either started by `python -m torch.distributed.run` (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when invoked plainly as `python bench.py --gpus N`, by this
file itself: the parent starts N fresh child interpreters BEFORE importing torch or touching the GPU and forwards rank
0's line.  The path shards by image (independent replicas, no data-path collective) -> weak scaling; barrier + device
sync on both sides of the timed region, MAX over ranks, rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ARCH_NAMES = {"dconv": "ResNet50-DConv", "duc": "ResNet50-DUC", "hrnet_w32": "HRNet-W32"}
FP32_MATRIX_PEAK_TFLOPS = 157.3
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md); the bf16 path is HBM/L2-bound long before this  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY 8d: >= 50)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed warm-up steps (SURVEY 8d: >= 10)")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--arch", default="dconv", choices=["dconv", "duc", "hrnet_w32"])
    ap.add_argument("--backbone", default="resnet50",
                    help="dconv / duc only: another factory of the reference's nets.pose_resnet_* (resnet18 ... resnet152, wide_resnet50_2, resnext50_32x4d, "
                         "resnext101_32x8d; nets/pose_resnet_dconv.py:282-368).  Not a BASELINE config: no tracked tile table (tuned on this GPU, untimed setup), "
                         "no cpu_baseline; the line's metric names the backbone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, default); gloo only to exercise the N > 1 code path on a single-GPU box")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="compute dtype of the network (BASELINE metric: f32)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "micro"],
                    help="infer = forward + GaussTaylor decode (BASELINE metric, default); train = fwd+bwd+Adam step (config 4, fp32)")
    ap.add_argument("--no-sync-bn", action="store_true", help="train mode, N > 1: per-rank BN statistics (the reference's DDP solver syncs them)")
    ap.add_argument("--sync-bn-latency-us", type=float, default=0.0,
                    help="train mode: every SyncBatchNorm message additionally costs this many microseconds (a delay kernel on a side stream the "
                         "consumer waits for); at N = 1 it switches the SyncBatchNorm path on with nothing to exchange: an emulation of what "
                         "the 104 small messages per step expose on xGMI, for boxes with one GPU")
    ap.add_argument("--bucket-mb", type=float, default=32.0, help="train mode, N > 1: gradient all-reduce bucket size")
    ap.add_argument("--single-stream", action="store_true", help="infer mode: issue independent branches (HRNet) on one stream")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (infer: forward + decode; train: PoseTrainer.capture - "
                    "forward, loss, backward, collectives, Adam and repack on their three streams)")
    ap.add_argument("--torch-collectives", action="store_true", help="train mode: issue the step's collectives through torch.distributed (each on RCCL's own "
                    "stream, fenced by events) instead of RCCL directly on the step's streams (sp_comm_allreduce_sum_f32); also where the emulated "
                    "--sync-bn-latency-us is spent")
    ap.add_argument("--preflight-rccl", action="store_true", help="train mode on ONE GPU: join a 1-rank RCCL group and issue every SyncBatchNorm / gradient "
                    "all-reduce of the step through it (nothing to exchange, but the calls, their streams and - with --graph - their capture are real)")
    ap.add_argument("--tiles", default=None, help="JSON tile table: loaded if it exists (skips autotune), else written.  Default: the tracked "
                    "table of this (arch, dtype) under profiles/ when the batch is 128 (the table the committed rocprofv3 summaries were taken with)")
    ap.add_argument("--no-train-autotune", action="store_true", help="train mode: keep the built-in tile heuristic")
    ap.add_argument("--retune", action="store_true", help="ignore the tracked tile table: time every (tile, kernel) per layer shape on this GPU")
    ap.add_argument("--layers-out", default=None, help="write the per-layer timing table (JSON) here")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline object)")
    ap.add_argument("--fuse-bottlenecks", action="store_true", help="(default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--pipeline-decode", action="store_true", help="infer mode: decode of step i on its own stream while the forward of step i + 1 runs "
                    "(engine.PipelinedForward; same results).  Default for hrnet_w32 (+4 %%: its forward ends in low-occupancy launches); "
                    "measured -2 %% on the ResNets in bf16, neutral in fp32")
    ap.add_argument("--interleave", type=int, default=None, help="infer mode: consecutive steps on this many independent streams, each with its own "
                    "activation pool (engine.InterleavedForward; same results).  Default 2 for the ResNets (two batches in flight fill the launch "
                    "boundaries and partial last rounds of each other's kernels: +2.7 %% fp32, +10 %% bf16), 3 for hrnet_w32 with every forward on "
                    "ONE stream (+20 %% over one forward spread over four branch streams)")
    ap.add_argument("--multi-stream", action="store_true", help="hrnet_w32 with --interleave > 1: keep the per-branch streams inside each forward")
    ap.add_argument("--no-pipeline-decode", action="store_true", help="hrnet_w32: decode in line, on the forward's stream")
    ap.add_argument("--fuse-blocks", action="store_true", help="hrnet_w32 bf16: 32-channel BasicBlocks as one launch each (sp_basic_block_c32; same bits) - the "
                    "default since round 6; kept for old command lines")
    ap.add_argument("--no-fuse-blocks", action="store_true", help="hrnet_w32 bf16: one launch per conv in the 32-channel branch (same-box A/Bs)")
    ap.add_argument("--fuse-blocks64", action="store_true", help="hrnet_w32 bf16: the 64-channel BasicBlocks as one launch each too (sp_basic_block_c64; same bits; "
                    "pays at small batch, profiles/r06_bb64_ab.txt)")
    ap.add_argument("--no-fuse-stem", action="store_true", help="infer mode: run the stem launch by launch (layout change, conv(s), pooling) instead of "
                    "as one launch (sp_stem7_pool for the ResNets, sp_hrnet_stem for HRNet in bf16; same bits)")
    ap.add_argument("--no-fuse-bottlenecks", action="store_true",
                    help="infer mode, bf16 ResNets: run the identity-shortcut Bottlenecks with 64 mid channels (layer1.1, layer1.2) conv by conv "
                         "instead of as one launch each (sp_bottleneck_c64; same bits)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="preflight of the N-rank job without touching a GPU: start the N ranks exactly as a real run does, rendezvous over gloo, "
                         "check the rank -> device mapping, one barrier and the MAX/SUM reductions of the measurement, print ONE JSON line; "
                         "non-zero exit when any rank dies")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (N = 1, infer, dconv f32 bs=128) only: skip the other BASELINE configs that are otherwise timed after the "
                         "headline with the same --steps / --warmup and appended as `other_configs` (DUC bf16 bs=128, HRNet-W32 bf16 bs=128, the "
                         "32-image bf16 train step)")
    ap.add_argument("--self-check-only", action="store_true",
                    help="rank job (N > 1): run the collective-path self-check (comm_select.self_check_rank: two small train steps through the native "
                         "RCCL path and through torch.distributed, compared bit for bit), every rank prints the agreed verdict as one JSON line; a "
                         "rank that cannot complete it exits 13.  Started by the supervisor (launch.run_job) under a deadline, never inside a training process")
    ap.add_argument("--no-extra-jobs", action="store_true",
                    help="N > 1, the driver's command (infer, dconv f32 bs=128): skip the two extra N-rank jobs that otherwise follow the headline "
                         "job - the collective self-check and the bf16 32-image-per-GPU train step (BASELINE config 4), appended as `other_configs`")
    ap.add_argument("--by-kernel", action="store_true", help="keep the per-instantiation table (`roofline.by_kernel`) in the JSON line")
    ap.add_argument("--native-comm", action="store_true", help="train mode, N > 1: the step's collectives through our own RCCL communicators on the "
                    "step's streams (sp_comm_*) instead of torch.distributed; opt-in until it has run on a multi-GPU box")
    ap.add_argument("--hsa-ipc-legacy", default="0", choices=["0", "1", "inherit"],
                    help="HSA_ENABLE_IPC_MODE_LEGACY for the ranks (recorded in config): 0 = dmabuf IPC, what this pool's host driver supports "
                         "(RCCL / tensor sharing across processes fails with hipIpcGetMemHandle otherwise); inherit = leave the environment alone")
    return ap.parse_args()


def apply_ipc_mode(args, env) -> str:
    """Set (or leave) HSA_ENABLE_IPC_MODE_LEGACY in `env` as --hsa-ipc-legacy says; returns what the ranks will see."""
    if args.hsa_ipc_legacy != "inherit":
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = args.hsa_ipc_legacy
    return env.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")


def dry_rank(args) -> int:
    """One rank of `bench.py --gpus N --dry-launch` (no GPU call): the same environment contract, rendezvous and reductions as a real
    rank, over gloo on the CPU."""
    import torch
    import torch.distributed as dist
    from simple_pose_amd.sharding import aggregate_throughput, rank_indices

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()                         # (counting devices does not initialise HIP)
    if world > 1:
        init_group("gloo", rank, world)
    if args.self_check_only or args.mode == "train":
        return dry_extra_job(args, rank, world)
    if os.environ.get("SP_BENCH_DRY_FAIL_RANK") == str(rank):    # test hook: a rank that dies after the rendezvous
        os._exit(3)
    device = local_rank % n_dev if n_dev else None
    mine = torch.tensor([rank, local_rank, -1 if device is None else device], dtype=torch.int64)
    table = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(table, mine)
        dist.barrier()
    else:
        table = [mine]
    # the measurement's own reductions on known numbers: SUM of the units, MAX of the elapsed time
    units, elapsed, value = aggregate_throughput(float(args.batch * args.steps), 1.0 + 0.01 * rank)
    ok = units == float(args.batch * args.steps * world) and abs(elapsed - (1.0 + 0.01 * (world - 1))) < 1e-9
    ranks = [int(t[0]) for t in table]
    devs = [int(t[2]) for t in table]
    ok = ok and ranks == list(range(world)) and [int(t[1]) for t in table] == list(range(world))
    one_per_device = n_dev >= world and devs == list(range(world))
    covered = sorted(i for r in range(world) for i in rank_indices(args.batch * world, r, world))
    ok = ok and covered == list(range(args.batch * world))      # DistributedSampler rule: every sample on exactly one rank
    # the train step's exchange plan (BASELINE config 4: ResNet50-DConv, 32 images per GPU, SyncBatchNorm as the reference's yaml has it),
    # computed on every rank from the model's own parameter table - host arithmetic only - and compared across the ranks
    from simple_pose_amd.nets import pose_resnet_dconv
    from simple_pose_amd.train import flat_layout, plan_gradient_buckets, sync_bn_messages_per_step
    net = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    offsets, numel = flat_layout([(n, tuple(p.shape)) for n, p in net.named_parameters()])
    buckets = plan_gradient_buckets(offsets, numel, args.bucket_mb)
    msgs = sync_bn_messages_per_step(list(net.state_dict().keys()))
    plan = {"gradient_buckets": len(buckets), "bucket_mbytes": [round(4 * (b["hi"] - b["lo"]) / (1 << 20), 1) for b in buckets],
            "gradient_floats": numel, "sync_bn_messages_per_step": msgs["per_step"], "batchnorm_layers": msgs["batchnorm_layers"],
            "communicators": 2 if args.native_comm else 1,
            "collective_path": "sp_comm_* (two RCCL communicators of our own)" if args.native_comm else "torch.distributed (nccl)"}
    sig = torch.tensor([len(buckets), msgs["per_step"], numel] + [b["lo"] for b in buckets], dtype=torch.int64)
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    if world > 1:
        dist.all_gather(sigs, sig)
    else:
        sigs = [sig]
    plan["same_on_every_rank"] = all(torch.equal(t, sig) for t in sigs)
    ok = ok and plan["same_on_every_rank"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(({"dry_launch": True, "train_step_plan": plan, "n_ranks": world, "rendezvous": "gloo, env:// on 127.0.0.1", "ranks_seen": ranks,
                          "visible_devices": n_dev, "rank_to_device": devs, "one_device_per_rank": one_per_device,
                          "reductions_ok": ok, "config": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")}}))
    return 0 if ok else 1


def dry_extra_job(args, rank: int, world: int) -> int:
    """The two extra N-rank jobs of the driver's command, without a GPU (gloo): the SAME decision code with the comparison's local verdict
    injected - SP_BENCH_TEST_SELF_CHECK = pass | fail | raise | hang (rank 1 is the odd one out; unset: the real answer over gloo, "not
    nccl") - and a train job that only agrees on the flag it was handed and reports the path it would take."""
    import torch.distributed as dist

    from simple_pose_amd import comm_select
    hook = os.environ.get("SP_BENCH_TEST_SELF_CHECK", "")
    odd = rank == min(1, world - 1)
    if args.self_check_only:
        def check():
            if hook == "hang" and odd:
                time.sleep(3600.0)                             # code that has never met a peer fails by hanging
            if hook == "raise" and odd:
                raise RuntimeError("sp_comm_create failed (test hook)")
            return (not (hook == "fail" and odd)), f"rank {rank}: injected verdict ({hook or 'none'})"
        try:
            out = comm_select.decide(None, "nccl" if hook else "gloo", bool(hook), world, self_check=check,
                                     agree=lambda v: comm_select._agree_min(v, None))
        except comm_select.SelfCheckError as e:
            print(f"bench.py dry self-check: rank {rank} could not complete the comparison ({e}) - exit 13, no vote", file=sys.stderr, flush=True)
            os._exit(13)
        emit({"self_check_job": True, "dry_launch": True, "rank": rank, "world": world, "decision": out})
    else:
        mine = os.environ.get("SP_NATIVE_COMM") == "1"
        native = comm_select._agree_min(mine, None) if world > 1 else mine
        if rank == 0:
            emit({"dry_launch": True, "job": "train", "n_gpus": world, "native_flag_agreed": native, "dtype": args.dtype, "steps": args.steps, "warmup": args.warmup,
                  "collective_path": "sp_comm (RCCL on the step's streams)" if native else "torch.distributed",
                  "collective_self_check": handed_in_decision(), "config": {"global_batch": args.batch * world}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baseline(arch: str):
    """Oracle ("port" of the reference's CPU path: torch-CPU forward + C GaussTaylor decode) on the host cores,
    BASELINE configs[0]: bs=4, fp32, eval; ~10 s of CPU work."""
    import numpy as np
    import torch

    from oracle import nets_oracle, pose_oracle
    from simple_pose_amd import synth

    head = arch
    x = torch.from_numpy(synth.input_images(4, 0))
    tinv = synth.trans_inv_batch(4)
    if arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import hrnet_state_dict_shapes, load_cfg
        cfg = load_cfg(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"))
        sd = {k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(hrnet_state_dict_shapes(cfg, 17), 0).items()}
        fwd = lambda sd_, x_: nets_oracle.hrnet_forward(sd_, x_, cfg)
    else:
        sd = {k: torch.from_numpy(v) for k, v in
              synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), 0).items()}
        fwd = nets_oracle.FORWARDS["resnet50_" + head]
    threads = torch.get_num_threads()

    def one():
        with torch.no_grad():
            hm = fwd(sd, x)
        pose_oracle.decode_gauss_taylor(hm.numpy(), tinv)

    best = None
    for nthreads in sorted({threads, min(threads, 32), min(threads, 16)}, reverse=True):
        torch.set_num_threads(nthreads)   # bs=4 does not feed 128+ cores; report the best of a few thread counts
        for _ in range(2):
            one()
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if el > 6.0 or n >= 30:
                break
        if best is None or 4 * n / el > best[0]:
            best = (4 * n / el, nthreads, n)
    torch.set_num_threads(threads)
    rate, nthreads, n = best
    return {"value": round(rate, 2), "unit": "images/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} iterations of bs=4 {ARCH_NAMES[arch]} 256x192 fp32 forward (torch-CPU oracle) + C GaussTaylor decode, "
                      f"best of thread counts <= {threads} on {os.cpu_count()} logical CPUs"}


TRAIN_JOB_CONFIG = "ResNet50-DConv 256x192 bf16 train step (fwd+bwd+Adam), batch sharded 32 images per GPU over {n} ranks (BASELINE config 4: bs=256 on 8 GPUs)"


def wants_extra_jobs(args) -> bool:
    """N > 1 and the command the driver runs (the headline config, nothing overridden): the supervisor then also runs the collective
    self-check job and the N-rank train-step job, so that the first multi-GPU run measures BASELINE config 4 and proves (or safely
    rejects) the RCCL path."""
    return (args.gpus > 1 and args.mode == "infer" and args.arch == "dconv" and args.backbone == "resnet50" and args.dtype == "f32" and args.batch == 128 and
            not args.graph and not args.no_extra_jobs and not args.no_other_configs and not args.self_check_only)


def _deadline(name: str, default: float) -> float:
    return float(os.environ.get(f"SP_BENCH_{name}_DEADLINE_S", default))


def supervise(args, managed, world: int) -> int:
    """This process is a supervisor only: it has touched neither torch nor the GPU (a process that has initialised HIP must never be
    replaced or forked into ranks).  `managed` = the ranks whose processes it starts: all of them (`python bench.py --gpus N`, the
    launcher) or its own one (a worker of `python -m torch.distributed.run`, which then manages the ONE child of its rank: the N
    supervisors meet through files in a shared directory, simple_pose_amd/launch.py).  Every job = fresh interpreters of this same file
    with the env:// variables the reference's DDP solver reads (processors/ddp_pose_resnet_solver.py:36,85-93: rank doubles as the device
    index), a hard wall-clock deadline, and a kill of exactly the PIDs started here.  Jobs, in order:

      main         the command as given (the N inference replicas; or the train step when --mode train was asked for);
      self_check   N > 1 train steps only: comm_select's two-path comparison as a job of its own - a hang or a dead rank there costs the
                   deadline, not the run, and means torch.distributed for the train job;
      train        (the driver's command only) the bf16 32-image-per-GPU train step with the decision handed in, appended to the main
                   job's line as `other_configs`.

    Rank 0's supervisor prints the ONE JSON line; the exit code is the main job's."""
    from simple_pose_amd import launch

    launch.install_signal_handlers()                 # told to stop (torchrun after a failed worker, a driver's timeout): the rank processes go first
    per_rank = len(managed) < world
    me = managed[0] if per_rank else 0
    base = dict(os.environ)
    base.update({"MASTER_ADDR": base.get("MASTER_ADDR", "127.0.0.1") if per_rank else "127.0.0.1", "LOCAL_WORLD_SIZE": str(world)})
    apply_ipc_mode(args, base)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
    share_dir = None
    if per_rank:
        tag = "_".join(str(base.get(k, "x")) for k in ("MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
        share_dir = os.path.join(os.environ.get("SP_BENCH_SHARE_ROOT", "/tmp"), f"sp_bench_{os.getppid()}_{tag}")
        os.makedirs(share_dir, exist_ok=True)
        if me == 0:
            launch._CLEANUP_DIRS.append(share_dir)
    this = [sys.executable, os.path.abspath(__file__)]

    def job_env(name: str, extra=None) -> dict:
        e = dict(base)
        if per_rank:
            if name != "main":           # torchrun's store belongs to ITS job (the main one, run exactly as torchrun meant it): the others meet in a FileStore
                e["SP_BENCH_INIT_METHOD"] = "file://" + os.path.join(share_dir, f"{name}.store")
                e.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        else:
            e["MASTER_PORT"] = str(launch.free_port())
            e.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        e.update(extra or {})
        return e

    def run(name, argv, deadline, extra=None):
        return launch.run_job(name, this + argv, managed, world, job_env(name, extra), deadline, share_dir=share_dir, capture_rank=me)

    def agreed(check_job):
        """The self-check job's outcome as the decision every supervisor arrives at."""
        if not per_rank:
            return launch.collective_decision(check_job)
        launch.publish(share_dir, "decision", me, launch.collective_decision(check_job))
        return launch.collective_decision(check_job, launch.gather(share_dir, "decision", world, wait_s=60.0))

    common = (["--dist-backend", args.dist_backend, "--hsa-ipc-legacy", args.hsa_ipc_legacy] + (["--dry-launch"] if args.dry_launch else []) +
              (["--no-sync-bn"] if args.no_sync_bn else []) + ["--bucket-mb", str(args.bucket_mb)])
    check_argv = ["--gpus", str(world), "--self-check-only", "--mode", "train", "--dtype", "bf16", "--batch", "32"] + common
    auto_route = not (args.torch_collectives or args.native_comm or os.environ.get("SP_NATIVE_COMM") in ("0", "1"))

    main_extra = {}
    if args.mode == "train" and auto_route and not args.self_check_only and not args.dry_launch:
        # a train job asked for by hand: prove (or reject) the native path first, in a job of its own, then hand the verdict in
        dec = agreed(run("self_check", check_argv, _deadline("CHECK", 300.0)))
        main_extra = {"SP_NATIVE_COMM": "1" if dec["native"] else "0", "SP_COLLECTIVE_DECISION": json.dumps(dec)}
    job = run("main", sys.argv[1:], _deadline("MAIN", 1800.0), main_extra)
    line = launch.last_json(job)
    rc = 0 if job["status"] == "ok" else (job["rc"] or 1)
    if job["status"] != "ok":
        print(f"bench.py supervisor: the main job {job['status']}: {job['detail']}", file=sys.stderr)
    elif me == 0 and (line is None or len(job["lines"]) != 1):
        print(f"bench.py supervisor: expected ONE JSON line from rank 0, got {len(job['lines'])}", file=sys.stderr)
        rc = 1
    if rc == 0 and wants_extra_jobs(args):
        dec = agreed(run("self_check", check_argv, _deadline("CHECK", 300.0)))
        train_argv = ["--gpus", str(world), "--mode", "train", "--dtype", "bf16", "--batch", "32", "--steps", str(args.steps), "--warmup", str(args.warmup),
                      "--no-cpu-baseline", "--no-other-configs"] + common
        tjob = run("train", train_argv, _deadline("TRAIN", 420.0),
                   {"SP_NATIVE_COMM": "1" if dec["native"] else "0", "SP_COLLECTIVE_DECISION": json.dumps(dec)})
        rec = {"config": TRAIN_JOB_CONFIG.format(n=world), "n_gpus": world, "command": "python3 bench.py " + " ".join(train_argv),
               "job": {"status": tjob["status"], "wall_s": tjob["wall_s"]}, "collective_self_check": dec}
        tl = launch.last_json(tjob)
        if tjob["status"] == "ok" and tl is not None:
            for k in ("value", "unit", "ms_per_step", "dtype", "steps", "warmup", "step_split_ms", "collective_path", "collectives_per_step",
                      "host_enqueue_ms_per_step", "network_frac_of_matrix_peak", "rccl_census", "dry_launch", "native_flag_agreed"):
                if k in tl:
                    rec[k] = tl[k]
            rec["global_batch"] = (tl.get("config") or {}).get("global_batch")
            rec["rccl"] = (tl.get("config") or {}).get("rccl")
            rec["roofline_frac"] = (tl.get("roofline") or {}).get("frac")
        else:
            rec["error"] = f"train job {tjob['status']}: {tjob['detail'] or 'no JSON line from rank 0'}"
        if line is not None:
            line["other_configs"] = [rec]                      # LAST key of the one line
    if me == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if per_rank:
        launch.publish(share_dir, "done", me, {"rc": rc})
        if me == 0:                                            # rank 0 tidies up once every supervisor has said it is done (bounded wait)
            launch.gather(share_dir, "done", world, wait_s=30.0)
            import shutil
            shutil.rmtree(share_dir, ignore_errors=True)
    return rc


def init_group(backend: str, rank: int, world: int, dev_index=None) -> None:
    """The job's process group: env:// on 127.0.0.1 (RANK / WORLD_SIZE / MASTER_* as the reference's solver reads them), or the init method
    the supervisor handed in (SP_BENCH_INIT_METHOD: a FileStore for the extra jobs under torchrun, whose own store belongs to its job)."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"init_method": os.environ["SP_BENCH_INIT_METHOD"]} if os.environ.get("SP_BENCH_INIT_METHOD") else {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", dev_index)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)


def handed_in_decision():
    """The self-check job's verdict as the supervisor handed it to this rank (SP_COLLECTIVE_DECISION), or None."""
    try:
        return json.loads(os.environ["SP_COLLECTIVE_DECISION"])
    except (KeyError, ValueError):
        return None


def self_check_rank_job(args, rank: int, world: int, dev_index: int) -> int:
    """One rank of `bench.py --gpus N --self-check-only` on the GPU: the model of the train job on this rank's device, then
    comm_select.self_check_rank (which exits 13 itself when the comparison cannot complete)."""
    import torch
    import torch.distributed as dist

    from simple_pose_amd import _lib, comm_select, synth
    from simple_pose_amd.nets import pose_resnet_dconv

    if world < 2:
        raise SystemExit("--self-check-only is a rank job of an N > 1 run")
    _lib.lib()
    torch.cuda.set_device(dev_index)
    init_group(args.dist_backend, rank, world, dev_index)
    dev = torch.device("cuda", dev_index)
    model = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.conditioned_state_dict(layout, seed=0).items()}, strict=True)
    model = model.to(dev).train()
    rc = comm_select.self_check_rank(model, "bf16" if args.dtype == "bf16" else "fp32", not args.no_sync_bn, emit=emit)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    return rc


_JSON_FD = None


def claim_stdout() -> None:
    """The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner at communicator creation goes to the C
    stdout): from here on file descriptor 1 IS stderr for everything in this process, and only `emit` writes to the real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj) -> None:
    data = (json.dumps(obj) + "\n").encode()
    fd = _JSON_FD if _JSON_FD is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main():
    args = parse()
    if args.gpus > 1 and os.environ.get("SP_BENCH_CHILD") != "1":     # before torch is imported or the GPU is touched
        if "WORLD_SIZE" not in os.environ:
            raise SystemExit(supervise(args, list(range(args.gpus)), args.gpus))       # `python bench.py --gpus N`: the launcher of all N ranks
        if wants_extra_jobs(args) and int(os.environ["WORLD_SIZE"]) == args.gpus:
            # a worker of `python -m torch.distributed.run` given the driver's command: it becomes the supervisor of its own rank, so that
            # the self-check and the train-step job run here too, each with a deadline (any other command runs in this process, as given)
            raise SystemExit(supervise(args, [int(os.environ.get("RANK", "0"))], args.gpus))
    claim_stdout()
    ipc_mode = apply_ipc_mode(args, os.environ)         # before the first HIP call (default 0: dmabuf IPC, the only mode this pool's driver has)
    if args.dry_launch:
        raise SystemExit(dry_rank(args))
    others = measure_other_configs(args) if is_default_command(args) else None     # child processes: before torch / the GPU is touched here
    if args.mode == "micro":                           # the HBM-bound kernels of the path one by one (tools/bench_micro.py)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_micro
        raise SystemExit(bench_micro.main([], emit=emit))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path")
    if world > n_dev and args.dist_backend == "nccl":
        raise SystemExit(f"--gpus {world} on a node with {n_dev} GPU(s): RCCL needs one device per rank "
                         "(--dist-backend gloo shares devices, for exercising the N > 1 path only)")
    dev_index = local_rank % n_dev
    if args.self_check_only:
        raise SystemExit(self_check_rank_job(args, rank, world, dev_index))
    if world == 1 and args.preflight_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 2000))
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
    if world > 1:
        torch.cuda.set_device(dev_index)
        init_group(args.dist_backend, rank, world, dev_index)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where the measurement's scalar reductions live
    ctx = {"rank": rank, "world": world, "dev": dev, "red_dev": red_dev, "ipc_mode": ipc_mode}
    line = run_once(args, ctx)
    if others is not None and line is not None:
        line["other_configs"] = others                    # LAST key of the one JSON line
    abandoned = False
    if dist.is_initialized() and dist.get_backend() == "nccl":
        # N > 1 over RCCL (or the 1-rank preflight group): ask RCCL itself who is in the job - AFTER the measurement, on a helper thread
        # with a deadline: a census that never answers must not cost the line (the process then leaves without tearing the group down)
        import threading
        box = {}

        def work():
            try:
                torch.cuda.set_device(dev)
                box["rccl"] = rccl_census(dev, rank, world)
            except Exception as e:
                box["rccl"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(120.0)
        abandoned = th.is_alive()
        info = {"error": "no answer from the communicator census within 120 s (abandoned; the measurement above is complete)"} if abandoned else box.get("rccl")
        if line is not None and info is not None:
            cfg = dict(line["config"])
            cfg["rccl"] = info
            line["config"] = cfg
            if "other_configs" in line:                  # keep it the last key
                line["other_configs"] = line.pop("other_configs")
    if abandoned and line is not None:
        line["rccl_census"] = "abandoned"                 # top-level, so that a reader of the line cannot take the run for a checked one
        if "other_configs" in line:
            line["other_configs"] = line.pop("other_configs")
    if rank == 0 and line is not None:
        emit(line)
    if abandoned:
        # the measurement is complete and printed; the diagnostic that hung is named in the line (`rccl_census`, `config.rccl.error`) and on
        # stderr.  At N > 1 that is a failing exit code (4): a job whose ranks could not be counted must not read as a success
        # (SP_CENSUS_STRICT=0 turns that off; =1 turns it on for the 1-rank preflight group too).
        print(f"bench.py rank {rank}: RCCL communicator census abandoned after 120 s (line printed; see config.rccl.error)", file=sys.stderr)
        sys.stderr.flush()
        strict = os.environ.get("SP_CENSUS_STRICT", "1" if world > 1 else "0") == "1"
        os._exit(4 if strict else 0)
    if world > 1:
        dist.destroy_process_group()


# the other single-GPU BASELINE configs (BASELINE.json configs[2], [4] and the per-GPU shard of configs[3]) as command lines of this file
OTHER_CONFIGS = [
    {"config": "ResNet50-DUC 256x192 bs=128 bf16 forward+decode", "argv": ["--arch", "duc", "--dtype", "bf16"]},
    {"config": "HRNet-W32 256x192 bs=128 bf16 forward+decode", "argv": ["--arch", "hrnet_w32", "--dtype", "bf16"]},
    {"config": "ResNet50-DConv 256x192 bs=32/GPU bf16 train step (fwd+bwd+Adam), 1-GPU shard of the bs=256 DDP config",
     "argv": ["--mode", "train", "--dtype", "bf16", "--batch", "32"]},
    # the reference's SHIPPED arithmetic for that step (configs/ddp_fast_pose.yaml: `amp: False`, processors/ddp_pose_resnet_solver.py:110-133)
    {"config": "ResNet50-DConv 256x192 bs=32/GPU fp32 train step (fwd+bwd+Adam), the reference's shipped arithmetic (amp: False)",
     "argv": ["--mode", "train", "--dtype", "f32", "--batch", "32"]},
]


def is_default_command(args) -> bool:
    """The command the driver runs: N = 1, the headline config, nothing overridden."""
    return (args.gpus == 1 and os.environ.get("WORLD_SIZE", "1") == "1" and args.mode == "infer" and args.arch == "dconv" and args.dtype == "f32" and
            args.batch == 128 and not args.graph and args.tiles is None and not args.retune and args.interleave is None and
            not args.no_other_configs and not args.dry_launch and args.backbone == "resnet50")


def measure_other_configs(args):
    """The default command also times the other single-GPU BASELINE configs with the same --steps / --warmup and the tracked tile tables:
    each as a fresh interpreter of this file (its own process: its own memory pools, streams and clocks - run in the headline's process
    after it, DUC bf16 read 8 % low), one after the other, BEFORE this process imports torch or touches the GPU (a process that has
    initialised HIP must not start children).  Returns the compact records appended to the line as `other_configs`."""
    import subprocess
    env = dict(os.environ)
    apply_ipc_mode(args, env)
    out = []
    for oc in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__)] + oc["argv"] + ["--steps", str(args.steps), "--warmup", str(args.warmup),
                                                                        "--no-cpu-baseline", "--no-other-configs"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
            lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"exit code {r.returncode}: {(r.stderr or '').strip().splitlines()[-1:] or ['no output']}")
            ln = json.loads(lines[-1])
            rf = ln.get("roofline") or {}
            out.append({"config": oc["config"], "value": ln["value"], "unit": ln["unit"], "ms_per_step": ln["ms_per_step"], "dtype": ln["dtype"],
                        "steps": ln["steps"], "warmup": ln["warmup"], "roofline_frac": rf.get("frac"), "roofline_kernel": rf.get("kernel"),
                        "network_frac_of_matrix_peak": ln.get("network_frac_of_matrix_peak"),
                        "batches_in_flight": ln["config"].get("batches_in_flight", 1), "tile_table": ln["config"].get("tile_table"),
                        "command": "python3 bench.py " + " ".join(cmd[2:-1]), "wall_s": round(time.perf_counter() - t0, 1)})
        except Exception as e:       # the headline line must still go out; a failed extra config is reported, not hidden
            out.append({"config": oc["config"], "error": f"{type(e).__name__}: {e}"[:300]})
    return out


def run_once(args, ctx):
    """One measurement (warm-up, timed region, per-kernel events) of the configuration `args` describes, in this process; returns the
    JSON line as a dict on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist

    rank, world, dev, red_dev, ipc_mode = ctx["rank"], ctx["world"], ctx["dev"], ctx["red_dev"], ctx["ipc_mode"]
    from simple_pose_amd import _lib, synth
    from simple_pose_amd.metrics import GaussTaylorKeyPointDecoder
    from simple_pose_amd.nets import pose_resnet_dconv, pose_resnet_duc

    _lib.lib()  # fail loudly if the HIP library is missing
    if args.arch == "hrnet_w32":
        from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
        model = get_pose_net(os.path.join(ROOT, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
        sd = synth.conditioned_state_dict(hrnet_state_dict_shapes(model.cfg, 17), seed=0)
    else:
        mod = {"dconv": pose_resnet_dconv, "duc": pose_resnet_duc}[args.arch]
        if not hasattr(mod, args.backbone):
            raise SystemExit(f"--backbone {args.backbone}: no such factory in nets.pose_resnet_{args.arch}")
        model = getattr(mod, args.backbone)(pretrained=False, num_classes=17)
        # names / shapes / dtypes come from the model itself (its state_dict layout is the reference's, tests/test_host_logic.py)
        layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()]
        sd = synth.conditioned_state_dict(layout, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    if args.no_fuse_stem:
        model.fuse_stem = False
    if args.arch == "hrnet_w32":
        if args.fuse_blocks:
            model.fuse_blocks = True
        if args.no_fuse_blocks:
            model.fuse_blocks = False
        if args.fuse_blocks64:
            model.fuse_blocks64 = True
    if args.dtype == "bf16" and args.mode == "infer":
        model.compute_dtype = "bf16"
        if args.arch in ("dconv", "duc"):
            model.fuse_bottlenecks = not args.no_fuse_bottlenecks
    decoder = GaussTaylorKeyPointDecoder(kernel_size=11, num_joints=17)

    B = args.batch
    import numpy as np
    base = synth.input_images(8, seed=100 + rank)  # 8 distinct random images per rank, tiled to the batch
    x = torch.from_numpy(np.concatenate([base] * ((B + 7) // 8), 0)[:B]).to(dev)
    tinv = torch.from_numpy(synth.trans_inv_batch(B)).to(dev)
    if args.mode == "train":
        if args.arch not in ("dconv", "duc", "hrnet_w32"):
            raise SystemExit("--mode train lowers the ResNet50 DConv / DUC nets and HRNet-W32")
        from simple_pose_amd.commons.transforms import RefineSimpleTransform
        from simple_pose_amd.train import PoseTrainer
        model.train()
        # N > 1: which path the step's collectives take is decided here, before the timed region (untimed setup): over an nccl group the
        # native RCCL path is taken when a start-up self-check reproduces torch.distributed bit for bit on every rank (comm_select), else
        # torch.distributed with the reason in the line; --torch-collectives / --native-comm force either
        from simple_pose_amd import comm_select
        coll = comm_select.select(model, None, 256, 192, "bf16" if args.dtype == "bf16" else "fp32", not args.no_sync_bn,
                                  requested=False if args.torch_collectives else (True if (args.native_comm and world > 1) else None))
        trainer = PoseTrainer(model, lr=1e-3, dtype="bf16" if args.dtype == "bf16" else "fp32", sync_bn=not args.no_sync_bn,
                              bucket_mb=args.bucket_mb, sync_bn_latency_us=args.sync_bn_latency_us,
                              native_comm=coll["native"] if world > 1 else None,
                              sync_bn_inline=not args.torch_collectives)
        # untimed setup: the tile of every forward / dgrad launch.  Default: the tracked table of this dtype under profiles/ (the one the
        # committed rocprofv3 summaries were taken with: no tuner launches in a profiled run, same launches on every rank and box);
        # --retune / a missing table: timed on rank 0 at this batch and shared
        tiles_src = "built-in heuristic"
        tpath = args.tiles or (None if (args.retune or args.arch != "dconv" or args.backbone != "resnet50") else tracked_tiles("train", args.dtype))   # the tracked tables are ResNet50-DConv's
        if not args.no_train_autotune:
            if tpath and os.path.isfile(tpath) and B == 32:
                with open(tpath) as fh:
                    trainer.set_tiles(json.load(fh), B)
                tiles_src = os.path.relpath(tpath, ROOT)
            else:
                table = trainer.autotune_shared(B)
                tiles_src = "autotuned on rank 0 (untimed setup)"
                if args.tiles and rank == 0 and not os.path.isfile(args.tiles):    # never overwrite a table tuned at another batch
                    with open(args.tiles, "w") as fh:
                        json.dump(table, fh)
        joints = torch.from_numpy(synth.joints_batch(B, 17, seed=200 + rank)).to(dev)
        targets, mask = RefineSimpleTransform.get_heat_map(joints, 2.0, (48, 64))   # HIP encoder, on device
        prog = None
        if args.preflight_rccl and world == 1:
            trainer.force_collectives = True
            trainer.sync_bn = not args.no_sync_bn
            if not args.torch_collectives:
                trainer._open_native_comm()

        def eager_step():
            loss = trainer.step(x, targets, mask)      # fwd + loss + bwd + gradient all-reduce (N > 1) + Adam + repack
            return (loss,)
        step = eager_step
        graphed_step = None
        if args.graph:
            graphed_step = trainer.capture(x, targets, mask)      # untimed setup: three eager steps, then the capture

            def step():
                return (graphed_step.step(graphed_step.x, graphed_step.targets, graphed_step.mask),)   # the batch is resident in the static inputs
    else:
        prog = model.hip_program(x)
        prog.multi_stream = not args.single_stream
        # untimed setup: pin the fastest workgroup tile per layer shape (or reuse a saved table: profiling runs do, so that
        # the trial launches of the tuner stay out of the per-kernel statistics)
        tiles_src = "autotuned on this GPU (untimed setup)"
        pinned = tracked_tiles(args.arch, args.dtype) if (args.tiles is None and not args.retune and B == 128 and args.backbone == "resnet50") else None
        if args.tiles and os.path.isfile(args.tiles):
            with open(args.tiles) as fh:
                prog.set_tiles(json.load(fh), B)
            tiles_src = os.path.relpath(args.tiles, ROOT)
        elif pinned:
            # results do not depend on the tile (same reduction order everywhere): the table only moves speed, and the tracked one is the
            # one profiles/ was measured with - kernel names, per-kernel traffic and this line then describe the same launches
            with open(pinned) as fh:
                prog.set_tiles(json.load(fh), B)
            tiles_src = os.path.relpath(pinned, ROOT)
        else:
            tiles = prog.autotune(x, in_situ=True)     # top candidates re-timed inside the running forward (untimed setup)
            if args.tiles and rank == 0:
                with open(args.tiles, "w") as fh:
                    json.dump(tiles, fh)

        if args.graph:
            graphed = prog.capture(x, decoder, tinv)

            def step():
                _, kps, mv = graphed()                  # inputs already sit in the graph's static buffers (resident in HBM)
                return kps, mv
        else:
            if args.interleave is None:
                args.interleave = 3 if args.arch == "hrnet_w32" else 2
            if args.arch == "hrnet_w32" and args.interleave > 1 and not args.multi_stream:
                prog.multi_stream = False      # batches on three streams beat branches on four: 24.0 -> 29.0 k img/s (the two do not add up: 4 hardware queues)
            if args.interleave > 1:
                from simple_pose_amd.engine import InterleavedForward
                inter = InterleavedForward(prog, decoder, depth=args.interleave)   # consecutive steps on independent streams / activation pools

                def step():
                    return inter(x, tinv)
            elif args.pipeline_decode or (args.arch == "hrnet_w32" and not args.no_pipeline_decode):
                from simple_pose_amd.engine import PipelinedForward
                piped = PipelinedForward(prog, decoder)     # decode of step i on its own stream under the forward of step i + 1

                def step():
                    return piped(x, tinv)
            else:
                hm_buf = torch.empty((B,) + tuple(prog.out_shape), dtype=torch.float32, device=dev)   # steady state: one resident result buffer

                def step():
                    hm = prog.run(x, out=hm_buf)
                    return decoder(hm, tinv)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
    torch.cuda.synchronize()

    # ---- timed region -------------------------------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(args.steps):
            out = step()
    host_enqueue_ms = 1e3 * (time.perf_counter() - t0) / args.steps     # the host's share: close to ms_per_step = launch-bound
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    from simple_pose_amd.sharding import aggregate_throughput
    total_images, elapsed, value = aggregate_throughput(float(B * args.steps), elapsed, device=red_dev)  # SUM units / MAX time
    assert os.environ.get("SP_CONV_DEBUG") or torch.isfinite(out[0]).all()

    ms_per_step = 1e3 * elapsed / args.steps

    # ---- the same steps with ONE batch in flight on one stream (rank 0, untimed extra: the mode of rounds 1-2 and of the per-kernel events,
    #      kept in every line so that rounds stay comparable whatever the default number of batches in flight becomes) ----
    one_in_flight = None
    if prog is not None and not args.graph and (args.interleave or 1) > 1:
        keep_ms = prog.multi_stream
        prog.multi_stream = False
        hm1 = torch.empty((B,) + tuple(prog.out_shape), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for _ in range(max(2, args.warmup // 2)):
                o1 = decoder(prog.run(x, out=hm1), tinv)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                o1 = decoder(prog.run(x, out=hm1), tinv)
            torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        prog.multi_stream = keep_ms
        one_in_flight = {"value": round(B * args.steps / dt1, 1), "unit": "images/s", "ms_per_step": round(1e3 * dt1 / args.steps, 3),
                         "note": "this rank, one batch in flight, one stream, decode in line"}

    # ---- per-kernel roofline: HIP events around every conv launch, on the launch stream, same inputs ----
    roofline = None
    if rank == 0 and not args.no_kernel_events and prog is not None:
        roofline = kernel_roofline(prog, x, steps=max(3, min(args.steps, 10)), layers_out=args.layers_out,
                                   peak=FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS,
                                   arch=args.arch, dtype=args.dtype)

    if roofline is not None:
        roofline["mode"] = "one batch in flight, one stream (per-kernel HIP events; `value` above: config.batches_in_flight)"
        if not args.by_kernel:
            roofline.pop("by_kernel", None)                   # (kept in --layers-out / --by-kernel; profiles/*_kernel_stats.csv has the same table)

    step_split = None
    if args.mode == "train":                                  # untimed extra steps with phase events (every rank: collectives inside)
        if graphed_step is not None:                          # (timing events cannot live inside a graph: these steps run eagerly)
            graphed_step.release()
            step = eager_step
        trainer.profile = True
        acc = {}
        with torch.no_grad():
            for _ in range(5):
                step()
                for k, v in trainer.phase_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 5
        trainer.profile = False
        step_split = {k: round(v, 3) for k, v in acc.items()}
        if not args.no_kernel_events:      # every rank runs the extra steps (collectives inside); rank 0 reports its own events
            roofline = train_roofline(trainer, step, B, FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS,
                                      layers_out=args.layers_out)
    if rank == 0:
        name = ARCH_NAMES[args.arch]
        if args.backbone != "resnet50" and args.arch in ("dconv", "duc"):
            name = name.replace("ResNet50", args.backbone)
        if args.mode == "train":
            # train step ~ 3 x forward (fwd + dgrad + wgrad) = 32.56 GFLOP / image (HRNet: from the trainer's own layer table)
            gflop = 3 * (({"dconv": 10.8528, "duc": 11.7517}.get(args.arch) if args.backbone == "resnet50" else None) or sum(L.flops for L in trainer.layers.values()) / 1e9)
            line = {
                "metric": f"images/sec train step (fwd+bwd+Adam), {name} 256x192 bs={B}/GPU", "value": round(value, 1), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{name} 256x192 train step, bs={B} per GPU, {'fp32' if args.dtype == 'f32' else 'bf16 compute + fp32 master weights/Adam'}, train-mode BN (batch statistics), Adam lr 1e-3, "
                                       "targets from the HIP encoder; N > 1: SyncBatchNorm " + ("off" if args.no_sync_bn else "on") +
                                       f", gradients all-reduced in {args.bucket_mb:g} MB buckets overlapped with backward (RCCL)",
                           "images_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", "HSA_ENABLE_IPC_MODE_LEGACY": ipc_mode, "tile_table": tiles_src,
                           "collectives": ("RCCL (nccl backend)" if args.dist_backend == "nccl" else "gloo") if world > 1 else "none"},
                "gflop_per_image": round(gflop, 3), "network_tflops": round(value * gflop / 1e3, 2),
                "network_frac_of_matrix_peak": round(value * gflop / 1e3 / ((FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS) * world), 4),
                "roofline": roofline, "cpu_baseline": None, "final_loss": float(out[0].item()), "step_split_ms": step_split,
                "host_enqueue_ms_per_step": round(host_enqueue_ms, 3), "sync_bn_latency_us_emulated": args.sync_bn_latency_us,
                "step_as_hip_graph": bool(args.graph), "rccl_preflight_one_rank": bool(args.preflight_rccl and world == 1),
                "collective_path": "torch.distributed" if (args.torch_collectives or trainer._comm is None and world > 1) else ("sp_comm (RCCL on the step's streams)" if trainer._comm is not None else "none"),
                # N > 1: the verdict of the supervised self-check job when one was handed in (what decided), else this process's own decision
                "collective_self_check": handed_in_decision() or {k: coll[k] for k in ("path", "reason", "self_check")},
                "collectives_per_step": {"gradient_buckets": len(trainer.buckets) if world > 1 else 0, "sync_bn": trainer.collective_count}}
        else:
            peak = FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS
            line = {
                "metric": "images/sec fwd+decode, %s 256x192 bs=%d" % (name, B),
                "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{name} 256x192 bs={B} per GPU, {'fp32' if args.dtype == 'f32' else 'bf16 (fp32 accumulate)'} forward "
                                       "(NCHW in -> heat maps) + GaussTaylor decode, eval-mode BN, conditioned random weights",
                           "images_per_gpu": B, "global_batch": B * world, "parallelism": f"replicas x{world} (no collective)",
                           "ranks": f"{world} process(es), one per GPU" + (f", {args.dist_backend} for the barrier / MAX only" if world > 1 else ""),
                           "launch": "one hipGraph per step" if args.graph else "stream launches", "tile_table": tiles_src,
                           "batches_in_flight": 1 if args.graph else (args.interleave or 1),
                           "decode": "in line" if (args.graph or (args.interleave or 1) > 1) else ("own stream, under the next step's forward (engine.PipelinedForward)" if (args.pipeline_decode or (args.arch == "hrnet_w32" and not args.no_pipeline_decode)) else "in line"),
                           "HSA_ENABLE_IPC_MODE_LEGACY": ipc_mode},
                "gflop_per_image": round(prog.flops_per_image / 1e9, 4),
                "network_tflops": round(value * prog.flops_per_image / 1e12, 2),
                "network_frac_of_matrix_peak": round(value * prog.flops_per_image / 1e12 / (peak * world), 4),
                "roofline": roofline,
                "one_batch_in_flight": one_in_flight,
                "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
                "cpu_baseline": None if (args.no_cpu_baseline or world > 1 or args.backbone != "resnet50") else cpu_baseline(args.arch),   # rank 0 at N = 1 only
            }
    else:
        line = None
    # give this configuration's pools / streams back before the next one is built in the same process
    for obj in ("inter", "piped"):
        o = locals().get(obj)
        if o is not None and hasattr(o, "close"):
            o.close()
    if args.mode == "train":
        trainer.close()
    del step, out
    prog = trainer = model = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return line


def rccl_census(dev, rank: int, world: int):
    """What RCCL itself reports for a communicator of our own over the job's ranks (ncclCommCount / ncclCommUserRank / ncclCommCuDevice
    through sp_comm_info), gathered over all ranks: an N-GPU line then carries proof that N ranks on N devices met."""
    import ctypes

    import torch
    import torch.distributed as dist

    from simple_pose_amd import _lib
    lib = _lib.lib()
    if not lib.sp_comm_available():
        return {"error": "librccl not resolvable (sp_comm_available() == 0)"}
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (ctypes.c_ubyte * 128)()
        _lib.check(lib.sp_comm_unique_id(buf), "sp_comm_unique_id")
        idt = torch.tensor(list(buf), dtype=torch.uint8)
    idt = idt.to(dev)
    if world > 1:
        dist.broadcast(idt, src=0)
    comm = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(lib.sp_comm_create(bytes(idt.cpu().tolist()), world, rank, ctypes.byref(comm)), "sp_comm_create")
    n, r, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(lib.sp_comm_info(comm, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "sp_comm_info")
    # one real exchange through it: every rank contributes (rank + 1), the sum must be world (world + 1) / 2
    probe = torch.full((4,), float(rank + 1), dtype=torch.float32, device=dev)
    _lib.check(lib.sp_comm_allreduce_sum_f32(comm, _lib.ptr(probe), 4, _lib.current_stream(dev)), "census all-reduce")
    torch.cuda.synchronize(dev)
    mine = torch.tensor([n.value, r.value, d.value, int(round(float(probe[0].item())))], dtype=torch.int64, device=dev)
    table = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(table, mine)
    else:
        table = [mine]
    _lib.check(lib.sp_comm_destroy(comm), "sp_comm_destroy")
    rows = [[int(v) for v in t.cpu().tolist()] for t in table]
    return {"ncclCommCount": sorted({row[0] for row in rows}), "ranks": [row[1] for row in rows], "devices": [row[2] for row in rows],
            "allreduce_sum_of_rank_plus_1": sorted({row[3] for row in rows}), "expected_sum": world * (world + 1) // 2}


def train_roofline(trainer, step, B: int, peak: float, steps: int = 3, layers_out=None):
    """Train step: HIP events around every conv-family launch (forward, dgrad, wgrad), recorded on the stream the launch goes to
    (weight gradients run on the trainer's second stream), over `steps` untimed extra steps.  The weight gradients go out in groups:
    one `sp_conv2d_wgrad_batched` call = conv_wgrad_group_kernel (every layer of the group) + its fixed-order wgrad_fold_kernel, timed
    together (a launch's time = the sum of the two kernels in profiles/*_train_*_kernel_stats.csv); every group is priced against the
    MFMA peak with 2 x MACs of its layers as algorithmic FLOPs.  Kernels on the two streams overlap, so a group's time is the
    time its launches were resident, not exclusive use of the chip."""
    import torch

    trainer.kernel_events = []
    with torch.no_grad():
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    ev, trainer.kernel_events = trainer.kernel_events, None
    groups = {}
    per = {}
    for kind, name, flops, e0, e1 in ev:
        g = groups.setdefault(kind, [0.0, 0.0, 0])
        ms = e0.elapsed_time(e1)
        g[0] += ms; g[1] += flops * B; g[2] += 1
        q = per.setdefault((kind, name), [0.0, flops * B, 0])
        q[0] += ms; q[2] += 1
    if layers_out:
        with open(layers_out, "w") as fh:
            json.dump([{"kind": k, "layer": n, "us": round(1e3 * v[0] / v[2] * (v[2] / steps if k == "wgrad" else 1.0), 1), "gflop": round(v[1] / 1e9, 2),
                        "tflops": round(v[1] / (v[0] / v[2] * 1e-3) / 1e12, 1)} for (k, n), v in per.items()], fh, indent=0)
    names = {"wgrad": f"conv_wgrad_group_kernel<{'true' if trainer.bf16 else 'false'}> + wgrad_fold_kernel",
             "forward": "conv_igemm_kernel<...> (forward launches)", "dgrad": "conv_igemm_kernel<...> (dgrad launches)"}
    dom = max(groups, key=lambda k: groups[k][0])
    ms, fl, n = groups[dom]
    ach = fl / (ms * 1e-3) / 1e12
    traffic = None
    dt = "bf16" if trainer.bf16 else "f32"
    if dom == "wgrad":                                  # a group launch = the unit kernel + its fold: PMC bytes of both (profiles/rNN_train_*_traffic.json)
        parts = [lookup_traffic(k, "train", dt) for k in (f"conv_wgrad_group_kernel<{'true' if trainer.bf16 else 'false'}>", "wgrad_fold_kernel")]
        traffic = None if any(p is None for p in parts) else int(sum(parts))
    else:
        # a family of conv_igemm instantiations: forward launches carry the STATS epilogue (8th template flag), dgrad launches do not -
        # launch-weighted mean of the family's PMC bytes per launch
        traffic = family_traffic("train", dt, stats=(dom == "forward"))
    return {"bound": "mfma", "kernel": names[dom], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "launches_per_step": n // steps, "avg_launch_us": round(1e3 * ms / n, 2),
            "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
            "by_group": {names[k]: {"launches_per_step": g[2] // steps, "ms_per_step": round(g[0] / steps, 3), "tflops": round(g[1] / (g[0] * 1e-3) / 1e12, 1)}
                         for k, g in sorted(groups.items(), key=lambda kv: -kv[1][0])},
            "note": "streams overlap: group times are residency, their sum exceeds the step; BN / pointwise passes are HBM-bound and listed in profiles/*_train_*"}


def _variant_name(op):
    """Kernel instantiation a conv launch resolves to (the name rocprofv3 reports): asked of the library's own dispatch
    (sp_conv2d_kernel_name), so it cannot drift from what is launched."""
    from simple_pose_amd import _lib
    if op.kind == "bb32":
        return _lib.conv_kernel_name(op.desc, False, 4)
    if op.kind == "bb64":
        return _lib.conv_kernel_name(op.desc, False, 6)
    if op.kind == "bneck64":
        return _lib.conv_kernel_name(op.desc, False, 5)
    if op.kind == "dual1x1":
        return "dual_pw_bf16_kernel" if op.w.element_size() == 2 else "conv_pw_dual_kernel<64>"
    if op.kind == "htrans":
        return "hrnet_transition1_kernel"
    if op.kind == "hstem":
        return "hrnet_stem_kernel"
    if op.kind == "stem7":                        # conv1 + bn1 + relu + maxpool in one launch (its time includes the pooling)
        return "stem_pool_kernel<%s>" % ("true" if op.w.element_size() == 2 else "false")
    return _lib.conv_kernel_name(op.desc, op.res is not None, 3 if getattr(op, "direct", False) else 0)


def tracked_tiles(arch: str, dtype: str):
    """Newest profiles/rNN_<arch>_<dtype>_tiles.json (written by tools/run_profiles.sh: the autotuned table of the profiled runs)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{arch}_{dtype}_tiles.json")), reverse=True)
    return cands[0] if cands else None


def family_traffic(arch: str, dtype: str, stats: bool):
    """Launch-weighted mean HBM bytes per launch over the conv_igemm_kernel instantiations of the newest profiles/rNN_<arch>_<dtype>_traffic.json
    whose STATS template flag (the 8th) equals `stats`; None without a table."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{arch}_{dtype}_traffic.json")), reverse=True)
    if not cands:
        return None
    with open(cands[0]) as fh:
        tab = json.load(fh)
    num = den = 0.0
    for k, v in tab.items():
        m = re.match(r"conv_igemm_kernel<(.*)>", k)
        if not m:
            continue
        flags = [t.strip() for t in m.group(1).split(",")]
        if len(flags) >= 8 and (flags[7] == "true") == stats:
            num += v["hbm_bytes_per_launch"] * v["launches_profiled"]
            den += v["launches_profiled"]
    return int(num / den) if den else None


def lookup_traffic(kernel: str, arch: str, dtype: str):
    """HBM bytes per launch of `kernel` from the committed PMC summaries (profiles/rNN_<arch>_<dtype>_traffic.json, newest round
    first; FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 correction applied by tools/profile_summary.py).  A kernel the
    table does not know is reported on stderr, not swallowed: a stale table must show."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{arch}_{dtype}_traffic.json")), reverse=True)
    if arch == "dconv" and dtype == "f32":
        cands.append(os.path.join(ROOT, "profiles", "r01_traffic.json"))
    for path in cands:
        try:
            with open(path) as fh:
                tab = json.load(fh)
        except (OSError, ValueError):
            continue
        if kernel in tab:
            return tab[kernel].get("hbm_bytes_per_launch")
    print(f"bench.py: no PMC traffic entry for kernel `{kernel}` ({arch} {dtype}) under profiles/ - roofline.traffic is null "
          f"(re-run tools/run_profiles.sh at this commit)", file=sys.stderr)
    return None


def kernel_roofline(prog, x, steps: int, layers_out=None, peak=FP32_MATRIX_PEAK_TFLOPS, arch="dconv", dtype="f32"):
    """HIP events recorded on the launch stream around every conv launch of `steps` forward passes (same inputs as the
    timed region).  The dominant kernel = the conv_igemm instantiation with the largest total time; its achieved
    TFLOP/s = algorithmic FLOPs of its launches / their summed durations, against the fp32 matrix peak.  The
    rocprofv3 --kernel-trace --stats summary of this same command is committed under profiles/ (average duration of
    that kernel name must agree); HBM traffic per launch comes from the PMC passes summarised in profiles/*.json."""
    import torch

    from simple_pose_amd import _lib

    lib = _lib.lib()
    B = x.shape[0]
    bufs = dict(prog._alloc(B, x.device))
    bufs["input"] = x
    bufs[prog.out_name] = torch.empty((B,) + tuple(prog.out_shape), dtype=torch.float32, device=x.device)
    FLOP_KINDS = ("conv", "bb32", "bb64", "bneck64", "dual1x1", "stem7", "hstem", "htrans")
    conv_ops = [op for op in prog.ops if op.kind in FLOP_KINDS]            # every launch that carries algorithmic FLOPs
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in conv_ops]
          for _ in range(steps)]
    stream = _lib.current_stream()
    for s in range(steps):
        ci = 0
        for op in prog.ops:                      # the program's own dispatcher (one stream), events around the conv launches
            if op.kind in FLOP_KINDS:
                ev[s][ci][0].record()
                prog._launch(lib, op, bufs, B, stream)
                ev[s][ci][1].record()
                ci += 1
            else:
                prog._launch(lib, op, bufs, B, stream)
    torch.cuda.synchronize()
    per_layer = []
    for ci, op in enumerate(conv_ops):
        ms = sorted(ev[s][ci][0].elapsed_time(ev[s][ci][1]) for s in range(steps))[steps // 2]
        per_layer.append((op.name, ms, op.flops * B, _variant_name(op)))
    groups = {}
    for n, m, f, v in per_layer:
        g = groups.setdefault(v, [0.0, 0.0, 0])
        g[0] += m; g[1] += f; g[2] += 1
    dom = max(groups, key=lambda v: groups[v][0])
    d_ms, d_flop, d_n = groups[dom]
    tot_ms = sum(m for _, m, _, _ in per_layer)
    tot_flop = sum(f for _, _, f, _ in per_layer)
    achieved = d_flop / (d_ms * 1e-3) / 1e12
    if layers_out:
        with open(layers_out, "w") as fh:
            json.dump([{"layer": n, "us": round(1e3 * m, 1), "gflop": round(f / 1e9, 2), "tflops": round(f / (m * 1e-3) / 1e12, 1),
                        "kernel": v} for n, m, f, v in per_layer], fh, indent=0)
    traffic = lookup_traffic(dom, arch, dtype)
    # the runner-up by summed launch time (HRNet's 32- and 64-channel tile kernels trade places from box to box: both are named)
    second = sorted((v for v in groups if v != dom), key=lambda v: -groups[v][0])[:1]
    runner_up = None
    if second:
        g2 = groups[second[0]]
        runner_up = {"kernel": second[0], "launches_per_step": g2[2], "avg_launch_us": round(1e3 * g2[0] / g2[2], 2),
                     "frac": round(g2[1] / (g2[0] * 1e-3) / 1e12 / peak, 4), "share_of_conv_time": round(g2[0] / tot_ms, 3)}
    return {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "launches_per_step": d_n, "avg_launch_us": round(1e3 * d_ms / d_n, 2),
            "algorithmic_gflop_per_launch": round(d_flop / d_n / 1e9, 3),
            "share_of_conv_time": round(d_ms / tot_ms, 3), "runner_up": runner_up,
            "all_conv_kernels": {"launches_per_step": len(per_layer), "ms_per_step": round(tot_ms, 3),
                                 "achieved": round(tot_flop / (tot_ms * 1e-3) / 1e12, 2),
                                 "frac": round(tot_flop / (tot_ms * 1e-3) / 1e12 / peak, 4)},
            "by_kernel": {v: {"launches": g[2], "avg_us": round(1e3 * g[0] / g[2], 1), "tflops": round(g[1] / (g[0] * 1e-3) / 1e12, 1)}
                          for v, g in sorted(groups.items(), key=lambda kv: -kv[1][0])}}


if __name__ == "__main__":
    main()
