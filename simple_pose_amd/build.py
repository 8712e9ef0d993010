"""Build recipe of libsimple_pose_hip.so (gfx950 only; hipcc cross-compiles without a GPU).

    python -m simple_pose_amd.build [--force]          (SP_FORCE_BUILD=1 in the environment = --force)

The library is built IN-TREE (simple_pose_amd/lib/) so that it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsimple_pose_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


last_build = {"compiled": 0, "reused": 0, "linked": False}      # what the last build() call did (evidence for the "does it build" check)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link the library.  `force` (or SP_FORCE_BUILD=1 in the environment) recompiles every
    object; otherwise objects and the library are reused when newer than every source / header."""
    force = force or os.environ.get("SP_FORCE_BUILD", "0") == "1"
    last_build.update(compiled=0, reused=len(sources()), linked=False)
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs, todo = [], []
    last_build.update(reused=0)
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(
                [os.path.getmtime(src)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")) +
                                           glob.glob(os.path.join(ROOT, "include", "*.h"))]):
            todo.append([HIPCC, "-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-c",
                         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", obj])
        else:
            last_build["reused"] += 1
        objs.append(obj)

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    if todo:                                   # independent translation units: a few hipcc processes side by side (SP_BUILD_JOBS, default 4)
        from concurrent.futures import ThreadPoolExecutor
        jobs = max(1, min(int(os.environ.get("SP_BUILD_JOBS", "4")), len(todo)))
        with ThreadPoolExecutor(jobs) as pool:
            list(pool.map(compile_one, todo))
        last_build["compiled"] += len(todo)
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    last_build["linked"] = True
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
