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
