"""N > 1 bookkeeping on CPU: gloo backend, world size 2 (the driver runs the real multi-GPU bench)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simple_pose_amd.sharding import aggregate_throughput, rank_indices


def test_rank_indices_match_distributed_sampler_semantics():
    from torch.utils.data import DistributedSampler

    class _DS(torch.utils.data.Dataset):
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
        def __getitem__(self, i): return i

    for n, world in [(10, 2), (11, 4), (257, 8), (5, 8)]:
        for rank in range(world):
            ref = list(DistributedSampler(_DS(n), num_replicas=world, rank=rank, shuffle=False))
            assert rank_indices(n, rank, world) == ref
    assert rank_indices(100, 1, 4, batch_size=8) == list(range(1, 100, 4))[:24]
    with pytest.raises(ValueError):
        rank_indices(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units, elapsed, rate = aggregate_throughput(128.0 * 10, 1.0 + rank, device="cpu")  # rank 1 is the slow one
        dist.barrier()
        out[rank] = (units, elapsed, rate)
    finally:
        dist.destroy_process_group()


def test_whole_job_throughput_is_sum_of_units_over_max_time():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res[0] == res[1] == (2560.0, 2.0, 1280.0)
    assert aggregate_throughput(10.0, 2.0) == (10.0, 2.0, 5.0)   # no process group -> local


def _grad_worker(rank, world, port, out):
    """DDP semantics of the train step's collective: SUM all-reduce of the flat gradient buffer, 1/world folded into the
    optimizer (simple_pose_amd.train.PoseTrainer.all_reduce_grads), on a tiny CPU module standing in for the net."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_pose_amd.train import FlatParams, PoseTrainer
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5), torch.nn.Conv2d(5, 2, 1))
        flat = FlatParams(m)
        names = [n for n, _ in m.named_parameters()]
        assert all(p.data_ptr() == flat.view(n).data_ptr() for n, p in m.named_parameters())      # params are views
        for n, p in m.named_parameters():
            p.grad.fill_(float(rank + 1) * (names.index(n) + 1))                                     # per-rank gradients
        shim = PoseTrainer.__new__(PoseTrainer)
        shim.flat, shim.pg, shim.world = flat, None, dist.get_world_size()
        scale = PoseTrainer.all_reduce_grads(shim)
        out[rank] = (scale, [float(p.grad.flatten()[0]) for p in m.parameters()], flat.numel % 4)
    finally:
        dist.destroy_process_group()


def test_gradient_all_reduce_is_sum_with_scale_one_over_world():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_grad_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        scale, g0, pad = res[rank]
        assert scale == 0.5 and pad == 0
        assert g0 == [3.0 * (i + 1) for i in range(len(g0))]     # (1 + 2) * (index + 1): SUM over the two ranks


# ---------------------------------------------------------------------------------------------- which path the step's collectives take
def test_collective_path_decision_table():
    """simple_pose_amd.comm_select.decide without a process group: the static rows of the decision (round-4 verdict, next 4a)."""
    from simple_pose_amd.comm_select import decide

    assert decide(None, "nccl", True, 1)["path"] == "none"
    d = decide(None, "gloo", False, 2, self_check=lambda: (_ for _ in ()).throw(AssertionError("must not run over gloo")))
    assert d["path"] == "torch.distributed" and "gloo" in d["reason"] and d["self_check"] == "not run"
    assert decide(None, "nccl", False, 8)["path"] == "torch.distributed"                       # librccl missing
    assert decide(False, "nccl", True, 8)["path"] == "torch.distributed"                       # --torch-collectives
    d = decide(True, "nccl", True, 8)
    assert d["native"] and d["self_check"] == "not run"                                        # explicit request: taken as asked
    with pytest.raises(RuntimeError):
        decide(True, "gloo", False, 2)
    assert decide(None, "nccl", True, 2)["path"] == "torch.distributed"                        # automatic, but nothing to prove it with
    ok = decide(None, "nccl", True, 2, self_check=lambda: (True, "2 steps"), agree=lambda v: v)
    assert ok["path"] == "sp_comm" and ok["native"] and ok["self_check"].startswith("passed: 2 steps")
    bad = decide(None, "nccl", True, 2, self_check=lambda: (False, "parameters differ"), agree=lambda v: v)
    assert bad["path"] == "torch.distributed" and "parameters differ" in bad["self_check"]
    # a rank that cannot complete the comparison must NOT vote (its peers may be inside a collective: round-5 advisor finding): the
    # exception leaves as SelfCheckError and the agreement is never reached from the except path
    from simple_pose_amd.comm_select import SelfCheckError
    votes = []
    with pytest.raises(SelfCheckError, match="ZeroDivisionError"):
        decide(None, "nccl", True, 2, self_check=lambda: 1 / 0, agree=lambda v: votes.append(v) or v)
    assert votes == []


def test_self_check_job_outcome_decides_the_train_jobs_path():
    """simple_pose_amd.launch.collective_decision: the supervised self-check JOB's outcome as the decision (pure).  Native only when the job
    ended in time with exit code 0 everywhere and its agreed verdict says so; a hang or a dead rank means torch.distributed, with the reason."""
    import json
    from simple_pose_amd.launch import collective_decision

    yes = {"self_check_job": True, "rank": 0, "decision": {"path": "sp_comm", "native": True, "reason": "start-up self-check passed on every rank", "self_check": "passed: x"}}
    no = {"self_check_job": True, "rank": 0, "decision": {"path": "torch.distributed", "native": False, "reason": "did not reproduce", "self_check": "failed"}}
    job = lambda status, rec=None, detail="": {"name": "self_check", "status": status, "rc": 0 if status == "ok" else 1, "lines": [json.dumps(rec)] if rec else [],
                                              "detail": detail, "wall_s": 1.0}
    assert collective_decision(None)["path"] == "torch.distributed"
    d = collective_decision(job("ok", yes))
    assert d["native"] and d["path"] == "sp_comm" and d["job"] == {"status": "ok", "wall_s": 1.0}
    assert not collective_decision(job("ok", no))["native"]
    d = collective_decision(job("timeout", None, "no end within the 300 s deadline"))
    assert d["path"] == "torch.distributed" and "hung" in d["reason"] and "300 s" in d["reason"]
    d = collective_decision(job("died", yes, "rank 3 exited with code 13"))           # a verdict line from a job that then died counts for nothing
    assert d["path"] == "torch.distributed" and "rank 3" in d["reason"]
    assert collective_decision(job("ok"))["path"] == "torch.distributed"               # ended, but said nothing
    # per-rank supervisors (torchrun): native only if EVERY supervisor's view says so, and every supervisor reported
    v_yes, v_no = dict(yes["decision"]), dict(no["decision"])
    assert collective_decision(job("ok", yes), [v_yes, v_yes])["native"]
    d = collective_decision(job("ok", yes), [v_yes, v_no])
    assert not d["native"] and d["reason"] == "did not reproduce"
    d = collective_decision(job("ok", yes), [v_yes, None])
    assert not d["native"] and "ranks [1]" in d["reason"]


def test_supervised_job_deadline_and_dead_rank():
    """simple_pose_amd.launch.run_job with stand-in rank programs: a job that outlives its deadline is ended (exactly the PIDs started here)
    and reported as "timeout"; a rank that exits non-zero ends the job as "died"; a clean job hands back rank 0's JSON lines."""
    import sys
    import time
    from simple_pose_amd.launch import last_json, run_job

    env = dict(os.environ)
    prog = ("import os, sys, time, json\n"
            "r = int(os.environ['RANK'])\n"
            "mode = sys.argv[1]\n"
            "if mode == 'hang' and r == 1: time.sleep(600)\n"
            "if mode == 'die' and r == 1: sys.exit(7)\n"
            "if mode == 'die': time.sleep(600)\n"
            "print('not json')\n"
            "print(json.dumps({'rank': r, 'world': int(os.environ['WORLD_SIZE']), 'child': os.environ['SP_BENCH_CHILD']}))\n")
    ok = run_job("t_ok", [sys.executable, "-c", prog, "ok"], [0, 1, 2], 3, env, 60.0)
    assert ok["status"] == "ok" and ok["rc"] == 0 and last_json(ok) == {"rank": 0, "world": 3, "child": "1"} and len(ok["lines"]) == 1
    t0 = time.time()
    hung = run_job("t_hang", [sys.executable, "-c", prog, "hang"], [0, 1, 2], 3, env, 3.0)
    assert hung["status"] == "timeout" and "[1]" in hung["detail"] and time.time() - t0 < 30.0
    t0 = time.time()
    dead = run_job("t_die", [sys.executable, "-c", prog, "die"], [0, 1, 2], 3, env, 60.0, grace_s=0.5)
    assert dead["status"] == "died" and dead["rc"] == 7 and "rank 1" in dead["detail"] and time.time() - t0 < 30.0


def _decide_worker(rank, world, port, out):
    """Two ranks over gloo standing in for an nccl job: the agreement is the real all-reduce(MIN) over the group; the self-check's local
    verdict is injected (the comparison itself needs two GPUs: tests/test_gpu_train.py::test_two_rank_rccl_*)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_pose_amd.comm_select import _agree_min, decide
        agree = lambda v: _agree_min(v, None)
        # (a) every rank passes -> native on every rank
        a = decide(None, "nccl", True, world, self_check=lambda: (True, f"rank {rank} ok"), agree=agree)
        # (b) rank 1 alone sees a mismatch -> BOTH ranks fall back, each saying why
        b = decide(None, "nccl", True, world, self_check=lambda: (rank != 1, "exp_avg differ" if rank == 1 else "ok"), agree=agree)
        # (c) a comparison that RAISES is not an in-process case any more: the raising rank leaves without voting and the supervisor ends
        #     the job (tests/test_host_logic.py::test_bench_dry_launch_runs_the_three_jobs_with_deadlines, hook "raise")
        c = None
        # (d) the real group's backend: gloo -> torch.distributed without running anything
        d = decide(None, dist.get_backend(), False, world, self_check=lambda: (True, "never"), agree=agree)
        out[rank] = (a, b, c, d)
    finally:
        dist.destroy_process_group()


def test_collective_path_self_check_is_agreed_over_the_group():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_decide_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        a, b, c, d = res[rank]
        assert a["path"] == "sp_comm" and a["native"]
        assert b["path"] == "torch.distributed" and not b["native"]
        assert d["path"] == "torch.distributed" and d["self_check"] == "not run"
    assert "exp_avg differ" in res[1][1]["self_check"] and "another rank" in res[0][1]["self_check"]


def _env_flag_worker(rank, world, port, out):
    """comm_select.select() with the self-check job's verdict handed in as SP_NATIVE_COMM: honoured only after the ranks agreed on it (MIN)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_pose_amd import comm_select
        res = []
        for flags in (("1", "0"), ("0", "0"), ("1", "1")):
            os.environ["SP_NATIVE_COMM"] = flags[rank]
            res.append(comm_select.select(None, None, requested=None))
        os.environ.pop("SP_NATIVE_COMM", None)
        res.append(comm_select.select(None, None, requested=None))          # no verdict handed in: nothing unproven runs inside a training process
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_handed_in_verdict_is_agreed_before_it_is_honoured():
    """Round 6: the supervised self-check job hands its verdict to the train job as SP_NATIVE_COMM.  A rank whose supervisor saw a different outcome must not
    open communicators the others do not: select() all-reduces the flag (MIN) first.  Over gloo the native path cannot exist at all, and says so."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_env_flag_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        mixed, zeros, ones, none = res[rank]
        assert mixed["path"] == "torch.distributed" and not mixed["native"]
        assert ("another rank" in mixed["reason"]) == (rank == 0) and ("SP_NATIVE_COMM=0" in mixed["reason"]) == (rank == 1)
        assert zeros["path"] == "torch.distributed" and "SP_NATIVE_COMM=0" in zeros["reason"]
        assert ones["path"] == "torch.distributed" and "cannot exist here" in ones["reason"] and "gloo" in ones["reason"]
        assert none["path"] == "torch.distributed" and none["self_check"] == "not run"


def test_supervisor_told_to_stop_takes_its_rank_processes_with_it(tmp_path):
    """torchrun ends its workers with SIGTERM when another worker failed (and a driver's timeout does the same): a supervisor
    (launch.install_signal_handlers) then ends exactly the rank processes it started before it leaves - on a GPU they would otherwise sit in a
    collective forever."""
    import signal
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = f"import time, os; open(r'{tmp_path}/child_' + os.environ['RANK'], 'w').write(str(os.getpid())); time.sleep(600)"
    code = (f"import sys, os; sys.path.insert(0, {root!r})\n"
            "from simple_pose_amd import launch\n"
            "launch.install_signal_handlers()\n"
            "print(os.getpid(), flush=True)\n"
            f"launch.run_job('t', [sys.executable, '-c', {child!r}], [0, 1], 2, dict(os.environ), 600.0)\n")
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    sup = int(p.stdout.readline())
    t_end = time.time() + 30
    while time.time() < t_end and not all(os.path.exists(tmp_path / f"child_{r}") and (tmp_path / f"child_{r}").read_text() for r in (0, 1)):
        time.sleep(0.1)
    kids = [int((tmp_path / f"child_{r}").read_text()) for r in (0, 1)]
    os.kill(sup, signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for k in kids:
        with pytest.raises(ProcessLookupError):
            os.kill(k, 0)
