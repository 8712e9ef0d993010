"""Which path the train step's collectives take when there is more than one rank - decided at start-up, proven before it is used.

The step has two ways to exchange (SyncBatchNorm messages and gradient buckets; the reference: SyncBatchNorm + DistributedDataParallel,
processors/ddp_pose_resnet_solver.py:36,89-93):

  * "sp_comm"            RCCL called directly on the step's own streams (csrc/comm.hip, two private communicators): one host call per
                         message, nothing between the message and its consumer but the stream's order;
  * "torch.distributed"  the process group's collectives (one communicator on RCCL's own stream, five stream / event calls per message
                         from Python: with SyncBatchNorm on the step is host-bound).

The native path is the design, but it has never met a second rank on this pool's 1-GPU boxes.  So nobody has to trust it: the comparison
that `tests/test_gpu_train.py::test_two_rank_rccl_*` runs - a few small steps through BOTH paths from identical state, parameters / Adam
moments / BatchNorm buffers compared bit for bit on every rank, the verdict agreed over the group - decides, and the native path is taken
only if it reproduced torch.distributed exactly.

Where the comparison runs (round 6): in a JOB OF ITS OWN - N fresh rank processes under a supervisor that has not touched the GPU
(`simple_pose_amd.launch.run_job`, from `bench.py --gpus N` or `python -m simple_pose_amd.comm_select --gpus N`) with a hard deadline.  Code
that has never met a peer fails by hanging, and a rank that raises while its peers are inside a collective cannot vote: so a rank that
cannot complete the comparison EXITS NON-ZERO (`SelfCheckError`; it never joins an agreement from an except path), the supervisor ends
the other ranks, and a job that died or outlived its deadline means torch.distributed, with the reason carried in bench.py's line
(`collective_self_check`).  The job's verdict reaches the real job as SP_NATIVE_COMM=0|1; `select()` only honours it after the ranks
agreed on it (MIN).  Without that variable a plain `select()` inside a rank takes torch.distributed: nothing unproven, and nothing
without a deadline, runs inside a training process (SP_SELF_CHECK_INPROCESS=1 restores the in-process check for debugging).
The decision logic is pure (no GPU): `decide()` and `launch.collective_decision()`; covered over gloo in tests/test_multi_rank_cpu.py
and tests/test_host_logic.py.
"""
from __future__ import annotations

import copy
import os
import sys
from typing import Callable, Optional, Tuple


class SelfCheckError(RuntimeError):
    """This rank could not complete the two-path comparison.  Peers may be inside a collective at this moment, so the rank must not vote:
    it leaves (non-zero exit in the self-check job) and the supervisor ends the job."""


def _agree_min(ok: bool, group) -> bool:
    """True only if EVERY rank of `group` says ok (all-reduce MIN over the process group, on the device its backend wants)."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def decide(requested: Optional[bool], backend: str, rccl_available: bool, world: int,
           self_check: Optional[Callable[[], Tuple[bool, str]]] = None, agree: Optional[Callable[[bool], bool]] = None) -> dict:
    """The decision, as data: {"path": "none" | "sp_comm" | "torch.distributed", "native": bool, "reason": str, "self_check": str}.

    requested   True: the caller insists on the native path (raises when it cannot exist); False: torch.distributed; None: automatic.
    backend     the process group's backend name ("nccl" is RCCL on ROCm); rccl_available: `sp_comm_available()`.
    self_check  () -> (this rank's verdict, detail): runs the two-path comparison; only called on the automatic route when a peer exists.
                It returns only after this rank's collectives have completed; if it raises, the exception propagates as SelfCheckError
                and NO agreement is attempted (peers may be mid-step: an all-reduce issued from the except path would pair with their
                gradient bucket or hang) - the caller is a rank of the supervised self-check job and exits non-zero.
    agree       (local verdict) -> global verdict (every rank must pass; default: all-reduce MIN over the default group).
    Every rank must call this with the same arguments at the same point (the self-check and the agreement are collectives)."""
    if world <= 1:
        return {"path": "none", "native": False, "reason": "one rank: nothing to exchange", "self_check": "not run"}
    usable = backend == "nccl" and bool(rccl_available)
    why_not = (f"process group backend is {backend!r}, not nccl" if backend != "nccl" else "librccl could not be loaded (sp_comm_available() == 0)")
    if requested is False:
        return {"path": "torch.distributed", "native": False, "reason": "requested (native_comm=False / --torch-collectives)", "self_check": "not run"}
    if requested is True:
        if not usable:
            raise RuntimeError(f"native collectives requested but unavailable: {why_not}")
        return {"path": "sp_comm", "native": True, "reason": "requested (native_comm=True / --native-comm / SP_NATIVE_COMM=1)", "self_check": "not run"}
    if not usable:
        return {"path": "torch.distributed", "native": False, "reason": why_not, "self_check": "not run"}
    if self_check is None:
        return {"path": "torch.distributed", "native": False, "reason": "no self-check available: the native path is taken only when proven", "self_check": "not run"}
    try:
        ok, detail = self_check()
    except SelfCheckError:
        raise
    except Exception as e:
        raise SelfCheckError(f"{type(e).__name__}: {e}"[:300]) from e
    all_ok = (agree or (lambda v: _agree_min(v, None)))(bool(ok))
    if all_ok:
        return {"path": "sp_comm", "native": True, "reason": "start-up self-check passed on every rank", "self_check": f"passed: {detail}"}
    verdict = f"failed on this rank: {detail}" if not ok else f"passed here ({detail}) but failed on another rank"
    return {"path": "torch.distributed", "native": False, "reason": "start-up self-check did not reproduce torch.distributed bit for bit", "self_check": verdict}


def two_path_self_check(model, group, in_h: int, in_w: int, dtype: str, sync_bn: bool, images: int = 4, steps: int = 2,
                        trainer_kwargs: Optional[dict] = None) -> Tuple[bool, str]:
    """`steps` train steps of `images` images per rank through the native path and through torch.distributed, from identical copies of
    `model` (which is left untouched): parameters, Adam moments and BatchNorm buffers must be equal bit for bit between the two paths on
    this rank, and equal to rank 0's (the ranks hold replicas).  Needs the GPU and an nccl group; returns (verdict of this rank, detail)."""
    import torch
    import torch.distributed as dist

    from . import synth
    from .commons.transforms import RefineSimpleTransform
    from .train import PoseTrainer

    dev = next(model.parameters()).device
    rank = dist.get_rank(group)
    x = torch.from_numpy(synth.input_images(images, seed=4100 + rank)).to(dev)
    if (in_h, in_w) != (256, 192):
        if in_h > 256 or in_w > 192:
            raise ValueError("self-check inputs are crops of the 256x192 synthetic images")
        x = x[:, :, :in_h, :in_w].contiguous()
    n_joints = int(model.final_layer.out_channels)           # (DDPProcessor builds the net with model_cfg["num_joints"])
    joints = torch.from_numpy(synth.joints_batch(images, n_joints, seed=4200 + rank)).to(dev)
    targets, mask = RefineSimpleTransform.get_heat_map(joints, 2.0, (in_w // 4, in_h // 4))
    if tuple(targets.shape) != (images, n_joints, in_h // 4, in_w // 4) or tuple(mask.shape) != (images, n_joints):
        raise ValueError(f"self-check targets {tuple(targets.shape)} / mask {tuple(mask.shape)} do not match {images} x {n_joints} joints")
    kw = dict(trainer_kwargs or {})
    results = []
    for native in (True, False):
        m = copy.deepcopy(model).train()
        tr = PoseTrainer(m, in_h=in_h, in_w=in_w, lr=1e-3, dtype=dtype, sync_bn=sync_bn, process_group=group, native_comm=native, **kw)
        try:
            assert (tr._comm is not None) == native          # (sp_comm_create is itself a collective: every rank has it or none does)
            losses = [float(tr.step(x, targets, mask).item()) for _ in range(steps)]
            torch.cuda.synchronize(dev)
            bufs = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in m.buffers()])
            results.append((losses, tr.flat.data.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), bufs, tr.collective_count))
        finally:
            tr.close()
            del tr, m
    a, b = results
    ok, detail = True, ""
    names = ("losses", "parameters", "exp_avg", "exp_avg_sq", "BatchNorm buffers")
    for name, u, v in zip(names, a[:5], b[:5]):
        same = (u == v) if isinstance(u, list) else torch.equal(u, v)
        if not same and ok:
            ok, detail = False, f"{name} differ between the native and the torch.distributed path after {steps} steps"
    # replicas: every rank must hold rank 0's parameters (checksum in fp64, MAX - MIN over the group == 0).  Every rank gets here whatever
    # its local verdict: the two all-reduces are collectives.
    s = a[1].to(torch.float64).sum().reshape(1)
    hi, lo = s.clone(), s.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    if ok and float((hi - lo).item()) != 0.0:
        ok, detail = False, "ranks disagree on the updated parameters (replicas diverged)"
    if ok:
        detail = f"{steps} steps x {images} images per rank, {a[5]} SyncBatchNorm messages per step: both paths bitwise equal, replicas equal"
    return ok, detail


def select(model, group=None, in_h: int = 256, in_w: int = 192, dtype: str = "fp32", sync_bn: bool = True, requested: Optional[bool] = None,
           log=None, run_self_check: Optional[bool] = None) -> dict:
    """Decide for this job (collective call: every rank, at the same point, before the first step).

    requested None is the automatic route.  It honours SP_NATIVE_COMM=0|1 - the verdict of the supervised self-check JOB - after the ranks
    agreed on it (all-reduce MIN: a rank whose supervisor saw a different outcome must not open communicators the others do not).
    The two-path comparison itself runs here only when `run_self_check` is True (the rank entry of the self-check job, `self_check_rank`)
    or SP_SELF_CHECK_INPROCESS=1: it has no deadline of its own."""
    import torch.distributed as dist

    from . import _lib

    if not (dist.is_available() and dist.is_initialized()):
        return decide(requested, "none", False, 1)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    from_env = None
    if requested is None and world > 1 and os.environ.get("SP_NATIVE_COMM") in ("0", "1"):
        mine = os.environ["SP_NATIVE_COMM"] == "1"
        requested = _agree_min(mine, group)
        from_env = ("SP_NATIVE_COMM=1 on every rank" if requested else
                    ("SP_NATIVE_COMM=0" if not mine else "SP_NATIVE_COMM=1 here but 0 on another rank"))
        if requested and not (backend == "nccl" and _lib.lib().sp_comm_available()):
            requested, from_env = False, f"SP_NATIVE_COMM=1 but the native path cannot exist here (backend {backend!r})"
    if run_self_check is None:
        run_self_check = os.environ.get("SP_SELF_CHECK_INPROCESS", "0") == "1"
    check = (lambda: two_path_self_check(model, group, in_h, in_w, dtype, sync_bn)) if run_self_check else None
    out = decide(requested, backend, bool(_lib.lib().sp_comm_available()) if backend == "nccl" else False, world,
                 self_check=check, agree=lambda v: _agree_min(v, group))
    if from_env is not None:
        out["reason"] = f"{from_env} (the self-check job's verdict, agreed over the group)"
    elif check is None and out["reason"].startswith("no self-check available"):
        out["reason"] = ("not proven for this job: the two-path self-check runs as a supervised job of its own (bench.py --gpus N, or "
                         "python -m simple_pose_amd.comm_select --gpus N) and hands its verdict over as SP_NATIVE_COMM")
    if dist.get_rank(group) == 0:
        print(f"[simple_pose_amd] collectives: {out['path']} ({out['reason']}; self-check {out['self_check']})", file=log or sys.stderr, flush=True)
    return out


# ------------------------------------------------------------------------------------------------ the self-check as a job of its own
def self_check_rank(model, dtype: str = "bf16", sync_bn: bool = True, in_h: int = 256, in_w: int = 192, emit=None) -> int:
    """One rank of the self-check job (process group already initialised, `model` on this rank's device): run the comparison, agree, and
    have EVERY rank write the verdict as one JSON line (`emit`, default stdout) - each supervisor reads its own rank's line.
    A rank that cannot complete the comparison does not return: it reports on stderr and leaves with exit code 13, so that the
    supervisor ends the job (`launch.run_job`: status "died")."""
    import json
    import traceback

    import torch.distributed as dist

    rank = dist.get_rank()
    try:
        out = select(model, None, in_h, in_w, dtype, sync_bn, requested=None, run_self_check=True)
    except BaseException:
        traceback.print_exc()
        print(f"[simple_pose_amd] self-check: rank {rank} could not complete the comparison - leaving (exit 13), no vote", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(13)
    line = {"self_check_job": True, "rank": rank, "world": dist.get_world_size(), "decision": out}
    if emit is None:
        print(json.dumps(line), flush=True)
    else:
        emit(line)
    return 0


def run_self_check_job(n_ranks: int, argv: Optional[list] = None, deadline_s: float = 300.0, env: Optional[dict] = None) -> dict:
    """From a process that has NOT touched the GPU: start the self-check job (`n_ranks` fresh interpreters of bench.py --self-check-only,
    or `argv`), bounded by `deadline_s`, and turn its outcome into the decision (`launch.collective_decision`)."""
    from . import launch

    launch.install_signal_handlers()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = argv or [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_ranks), "--self-check-only", "--mode", "train", "--dtype", "bf16",
                    "--batch", "32"]
    e = dict(env if env is not None else os.environ)
    e.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(launch.free_port())})
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("SP_NATIVE_COMM", "TORCHELASTIC_USE_AGENT_STORE"):
        e.pop(k, None)
    job = launch.run_job("self_check", argv, list(range(n_ranks)), n_ranks, e, deadline_s)
    return launch.collective_decision(job)


if __name__ == "__main__":
    import argparse
    import json

    ap = argparse.ArgumentParser(description="Run the collective-path self-check as a supervised N-rank job and print its verdict "
                                             "(export SP_NATIVE_COMM accordingly before starting the training job).")
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--deadline", type=float, default=300.0)
    a = ap.parse_args()
    d = run_self_check_job(a.gpus, deadline_s=a.deadline)
    print(json.dumps(d))
    print(f"export SP_NATIVE_COMM={1 if d['native'] else 0}", file=sys.stderr)
