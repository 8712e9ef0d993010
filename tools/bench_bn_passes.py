"""The BatchNorm passes of the 32-image train step, one launch shape at a time (HIP events, operands rotated over enough buffer sets to
defeat the 256 MB Infinity Cache):  python tools/bench_bn_passes.py [--g16 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g16", type=int, default=1)
    ap.add_argument("--reps", type=int, default=24)
    args = ap.parse_args()
    from simple_pose_amd import _lib
    lib, P, dev, st = _lib.lib(), _lib.ptr, "cuda:0", _lib.current_stream()
    adt = torch.bfloat16
    gdt = torch.bfloat16 if args.g16 else torch.float32
    flag = 1 | (2 if args.g16 else 0)
    shapes = [("l1.bn1", 98304, 64, 768, False), ("l1.bn3", 98304, 256, 768, True), ("l2.bn1", 24576, 128, 192, False),
              ("l2.bn3", 24576, 512, 192, True), ("l3.bn1", 6144, 256, 96, False), ("l3.bn3", 6144, 1024, 96, True),
              ("l4.bn1", 1536, 512, 24, False), ("l4.bn3", 1536, 2048, 24, True), ("dc2", 98304, 256, 768, False)]
    for name, rows, C, prow, res in shapes:
        per = rows * C * 2
        nset = max(2, int(600e6 // (per * 4)) + 1)
        sets = []
        for _ in range(nset):
            z = torch.randn(rows, C, device=dev).to(adt)
            y = torch.relu(torch.randn(rows, C, device=dev)).to(adt)
            dy = torch.randn(rows, C, device=dev).to(gdt)
            r = torch.randn(rows, C, device=dev).to(adt) if res else None
            dres = torch.zeros(rows, C, device=dev, dtype=gdt) if res else None
            out = torch.empty(rows, C, device=dev, dtype=adt)
            sets.append((z, y, dy, r, dres, out))
        part = torch.randn(3, prow, C, device=dev)
        mean, invstd, gamma, beta = (torch.randn(C, device=dev) for _ in range(4))
        invstd = invstd.abs() + 0.5
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)

        def fwd(i):
            z, y, dy, r, dres, out = sets[i % nset]
            _lib.check(lib.sp_bn_fold_apply_nhwc(P(z), 1, P(part[0]), P(part[1]), prow, C, rows, 1e-5, 0.1, P(gamma), P(beta), P(r), P(out), rows, C, 1,
                                                 P(mean), P(invstd), P(rm), P(rv), None, st), name)

        def bwd(i):
            z, y, dy, r, dres, out = sets[i % nset]
            _lib.check(lib.sp_bn_fold_bwd_apply_nhwc(P(dy), flag, P(y), P(z), P(part[0]), P(part[1]), None, prow, C, P(mean), P(invstd), P(gamma), rows,
                                                     rows, C, P(dg), P(db), None, None, P(out), P(dres), 0, st), name)
        def fwd_plain(i):
            z, y, dy, r, dres, out = sets[i % nset]
            _lib.check(lib.sp_bn_apply_nhwc(P(z), 1, P(mean), P(invstd), P(gamma), P(beta), P(r), P(out), rows, C, 1, None, st), name)

        def bwd_plain(i):
            z, y, dy, r, dres, out = sets[i % nset]
            _lib.check(lib.sp_bn_train_bwd_apply_nhwc(P(dy), flag, P(y), P(z), P(mean), P(invstd), P(gamma), P(dg), P(db), rows, rows, C, P(out), P(dres), 0,
                                                      st), name)
        gb = 2 if args.g16 else 4
        fb, bb = rows * C * (4 + (2 if res else 0)), rows * C * (gb + 6 + (gb if res else 0))
        for kind, fn, nbytes in (("fwd", fwd, fb), ("fwd_plain", fwd_plain, fb), ("bwd", bwd, bb), ("bwd_plain", bwd_plain, bb)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.reps):
                fn(i)
            e1.record()
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
            print(f"{name:8s} {kind:9s} rows {rows:6d} C {C:5d}  {nbytes / 1e6:7.1f} MB {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
