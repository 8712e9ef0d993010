"""Supervision of N-rank jobs from a process that has touched neither torch nor the GPU (standard library only).

The reference starts its DDP solver with `python -m torch.distributed.launch ... main.py` and reads RANK / WORLD_SIZE / MASTER_* from the
environment (processors/ddp_pose_resnet_solver.py:36,85-93).  bench.py measures more than one such job per invocation at N > 1 (the
inference replicas, the collective self-check, the batch-sharded train step): each job is N fresh interpreters with that env:// contract,
run under a hard wall-clock deadline, and ended by killing exactly the PIDs started here.  Two shapes of supervisor share this code:

  * launcher  (`python bench.py --gpus N`): ONE parent manages all N ranks of every job;
  * per rank  (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): each of torchrun's N workers manages the ONE
    child of its own rank; the N supervisors see each other only through small files in a shared directory (`share_dir`: exit codes per
    job and rank, the agreed decision), and their children rendezvous through a FileStore there (torchrun's own store belongs to its job).

Nothing in here may import torch: a supervisor that initialised HIP could not start children safely.
"""
from __future__ import annotations

import json
import os
import selectors
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    """A free TCP port on the loopback interface (the rendezvous address is always 127.0.0.1: the container's hostname may not resolve)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rc_path(share_dir: str, name: str, rank: int) -> str:
    return os.path.join(share_dir, f"{name}.rank{rank}.rc")


def _write_atomic(path: str, text: str) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


_ACTIVE: List[subprocess.Popen] = []          # children of the job that is running now (for the signal handler)
_CLEANUP_DIRS: List[str] = []


def install_signal_handlers() -> None:
    """A supervisor that is told to stop (torchrun ends its workers with SIGTERM when another worker failed; a driver's timeout does the same) must not
    leave its rank processes behind - on a GPU they would sit in a collective forever: SIGTERM / SIGINT end exactly the children started here, then exit."""
    import shutil
    import signal

    def handler(signum, _frame):
        stop(list(_ACTIVE), wait_s=5.0)
        for d in _CLEANUP_DIRS:
            shutil.rmtree(d, ignore_errors=True)
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, handler)
        except (ValueError, OSError):          # (not the main thread)
            pass


def stop(procs: Sequence[subprocess.Popen], wait_s: float = 10.0) -> None:
    """End exactly the children in `procs` (SIGTERM, then SIGKILL for whatever ignores it)."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.time() + wait_s
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except Exception:
            p.kill()
            try:
                p.wait(timeout=5.0)
            except Exception:
                pass


def run_job(name: str, argv: List[str], ranks: Sequence[int], world: int, env: Dict[str, str], deadline_s: float,
            share_dir: Optional[str] = None, grace_s: float = 5.0, capture_rank: int = 0) -> dict:
    """Run one N-rank job: start `argv` once per rank in `ranks` (RANK / LOCAL_RANK / WORLD_SIZE added to `env`), wait for all of them under
    a hard deadline, end exactly those PIDs if it passes or any rank (here, or - through `share_dir` - under another supervisor) fails.

    Returns {"name", "status": "ok" | "died" | "timeout", "rc", "lines": JSON-looking stdout lines of `capture_rank` (if managed here),
    "detail", "wall_s"}.  Never raises for a failing job; stdout of the other ranks and non-JSON lines go to this process's stderr."""
    t0 = time.time()
    procs: Dict[int, subprocess.Popen] = {}
    for r in ranks:
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), SP_BENCH_CHILD="1")
        procs[r] = subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == capture_rank else sys.stderr, stderr=sys.stderr)
        _ACTIVE.append(procs[r])
    out = b""
    sel = selectors.DefaultSelector()
    cap = procs.get(capture_rank)
    if cap is not None:
        os.set_blocking(cap.stdout.fileno(), False)
        sel.register(cap.stdout, selectors.EVENT_READ)
    status, rc, detail = "ok", 0, ""
    written = set()

    def drain() -> None:
        nonlocal out
        if cap is None:
            return
        while True:
            try:
                chunk = os.read(cap.stdout.fileno(), 65536)
            except (BlockingIOError, OSError):
                return
            if not chunk:
                return
            out += chunk

    try:
        while True:
            if cap is not None:
                sel.select(timeout=0.2)
                drain()
            else:
                time.sleep(0.2)
            codes = {r: p.poll() for r, p in procs.items()}
            if share_dir:
                for r, c in codes.items():                    # tell the other supervisors as soon as one of our ranks has ended
                    if c is not None and r not in written:
                        _write_atomic(_rc_path(share_dir, name, r), str(c))
                        written.add(r)
            bad = [(r, c) for r, c in codes.items() if c not in (None, 0)]
            if bad:
                status, rc, detail = "died", bad[0][1], f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if share_dir and len(procs) < world:
                peers_bad = [(r, v) for r in range(world) if r not in procs
                             for v in [_read(_rc_path(share_dir, name, r))] if v is not None and v.strip() not in ("0", "")]
                if peers_bad:
                    status, rc, detail = "died", 1, f"rank {peers_bad[0][0]} (another supervisor) ended with {peers_bad[0][1].strip()}"
                    break
            if all(c == 0 for c in codes.values()):
                if not share_dir or len(procs) == world:
                    break
                if all(_read(_rc_path(share_dir, name, r)) is not None for r in range(world)):
                    break                                      # every rank of the job has ended, everywhere
            if time.time() - t0 > deadline_s:
                alive = [r for r, c in codes.items() if c is None]
                status, rc = "timeout", 124
                detail = (f"no end within the {deadline_s:g} s deadline (ranks still running here: {alive})" if alive else
                          f"ranks under other supervisors did not report within the {deadline_s:g} s deadline")
                break
    finally:
        if status == "died":
            # the others usually follow within moments for the same reason (no GPU, a failed rendezvous): let them say so themselves
            t_end = time.time() + grace_s
            while time.time() < t_end and any(p.poll() is None for p in procs.values()):
                time.sleep(0.1)
        stop(list(procs.values()))
        for p in procs.values():
            if p in _ACTIVE:
                _ACTIVE.remove(p)
        drain()
        if cap is not None:
            try:
                sel.unregister(cap.stdout)
            except Exception:
                pass
            try:
                rest = cap.stdout.read()
                if rest:
                    out += rest
            except Exception:
                pass
            cap.stdout.close()
        if share_dir:
            for r, p in procs.items():
                if r not in written:
                    _write_atomic(_rc_path(share_dir, name, r), "timeout" if status == "timeout" and p.returncode != 0 else str(p.returncode))
    text = out.decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.lstrip().startswith("{")]
    for ln in text.splitlines():
        if ln not in lines and ln.strip():
            print(ln, file=sys.stderr)
    return {"name": name, "status": status, "rc": rc, "lines": lines, "detail": detail, "wall_s": round(time.time() - t0, 1)}


def last_json(job: dict) -> Optional[dict]:
    """The last parsable JSON line of a finished job's captured stdout, or None."""
    for ln in reversed(job.get("lines") or []):
        try:
            return json.loads(ln)
        except ValueError:
            continue
    return None


def publish(share_dir: str, key: str, rank: int, value: dict) -> None:
    """This supervisor's view of `key` (e.g. the self-check verdict its rank saw), for `gather` on every supervisor."""
    _write_atomic(os.path.join(share_dir, f"{key}.rank{rank}.json"), json.dumps(value))


def gather(share_dir: str, key: str, world: int, wait_s: float) -> List[Optional[dict]]:
    """Every supervisor's `publish`ed view of `key` (None for a rank that did not publish within `wait_s`)."""
    t_end = time.time() + wait_s
    while True:
        got = []
        for r in range(world):
            txt = _read(os.path.join(share_dir, f"{key}.rank{r}.json"))
            try:
                got.append(json.loads(txt) if txt else None)
            except ValueError:
                got.append(None)
        if all(g is not None for g in got) or time.time() > t_end:
            return got
        time.sleep(0.1)


def collective_decision(check_job: Optional[dict], views: Optional[List[Optional[dict]]] = None) -> dict:
    """Which path the train job's collectives take, from the self-check JOB's outcome (pure: no process, no GPU).

    check_job  the `run_job` record of the self-check job as THIS supervisor saw it; None = no self-check was run.
    views      per-rank supervisors only: what every supervisor published (`{"native": bool, ...}`); native only if ALL say so.
    The native path (RCCL on the step's own streams) is taken only when the job ended in time, with exit code 0 everywhere, and its
    agreed verdict says so; a job that hung or died means torch.distributed, with the reason (simple_pose_amd.comm_select)."""
    if check_job is None:
        return {"path": "torch.distributed", "native": False, "reason": "no self-check job was run", "self_check": "not run"}
    job = {"status": check_job["status"], "wall_s": check_job["wall_s"]}
    if check_job["status"] == "timeout":
        return {"path": "torch.distributed", "native": False, "reason": f"self-check job hung: {check_job['detail']}; its ranks were killed",
                "self_check": "no verdict (deadline)", "job": job}
    if check_job["status"] != "ok":
        return {"path": "torch.distributed", "native": False, "reason": f"self-check job died: {check_job['detail']}",
                "self_check": "no verdict (a rank failed)", "job": job}
    rec = last_json(check_job)
    dec = (rec or {}).get("decision") if views is None else None
    if views is not None:
        if any(v is None for v in views):
            missing = [r for r, v in enumerate(views) if v is None]
            return {"path": "torch.distributed", "native": False, "reason": f"no self-check verdict from the supervisors of ranks {missing}",
                    "self_check": "no verdict", "job": job}
        firsts = [v for v in views if not v.get("native")]
        dec = dict(firsts[0] if firsts else views[0])
    if not dec or "native" not in dec:
        return {"path": "torch.distributed", "native": False, "reason": "self-check job ended without a verdict line", "self_check": "no verdict", "job": job}
    dec = dict(dec)
    dec["native"] = bool(dec["native"])
    dec["path"] = "sp_comm" if dec["native"] else "torch.distributed"
    dec["job"] = job
    return dec
