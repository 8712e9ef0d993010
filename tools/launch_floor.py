"""What a dependent launch costs on this GPU: N back-to-back launches of a do-nothing kernel (sp_stream_delay_us(0): one wave, one s_memtime) on one
stream, HIP events around the batch; and the same for the smallest real kernels of the train step (a 24-row BatchNorm fold, a 3 MB pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simple_pose_amd import _lib
lib, P, st, dev = _lib.lib(), _lib.ptr, _lib.current_stream(), "cuda:0"


def timed(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print("empty kernel            %.2f us per launch" % timed(lambda: lib.sp_stream_delay_us(0.0, st)))
C, prow, rows = 512, 24, 1536
part = torch.randn(3, prow, C, device=dev)
mean, invstd, rm, rv = (torch.ones(C, device=dev) for _ in range(4))
print("24-row fold (stats)     %.2f us" % timed(lambda: lib.sp_bn_train_stats_from_conv(P(part[0]), P(part[1]), prow, C, rows, C, 1e-5, 0.1, P(mean), P(invstd), P(rm), P(rv), st)))
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
print("24-row fold (bwd sums)  %.2f us" % timed(lambda: lib.sp_bn_bwd_sums_from_conv(P(part[0]), P(part[1]), prow, C, C, P(dg), P(db), st)))
z = torch.randn(rows, C, device=dev).to(torch.bfloat16); y = torch.empty_like(z)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
print("3 MB plain pass         %.2f us" % timed(lambda: lib.sp_bn_apply_nhwc(P(z), 1, P(mean), P(invstd), P(g), P(b), None, P(y), rows, C, 1, None, st)))
