"""CPU, world_size 2 over gloo: the N>1 path of bench.py (image sharding + bucketed, hook-driven gradient all-reduce with
grad-less parameters excluded) reproduces the single-process gradient of the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.unused = nn.Linear(8, 8)  # never contributes: like the 30 discarded-gate parameters
        self.b = nn.Linear(16, 4)

    def forward(self, x):
        with torch.no_grad():
            self.unused(x)
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tamtr_amd.dist import GradReducer, init_from_env, shard_batch
    r, _, w = init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = Tiny()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(1))
    lo, hi = shard_batch(8, rank, world)
    red = GradReducer(model.named_parameters(), bucket_bytes=256, skip=lambda n: n.startswith('unused'))
    assert len(red.buckets) > 1 and red.n_params == 4
    for step in range(2):  # twice: buckets are reused
        red.prepare()
        model(x[lo:hi]).pow(2).sum().backward()
        red.finish()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sum_matches_single_process(tmp_path):
    out = str(tmp_path / 'g.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Tiny()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(1))
    model(x).pow(2).sum().backward()
    assert set(got) == {'a.weight', 'a.bias', 'b.weight', 'b.bias'}
    for k, p in model.named_parameters():
        if k in got:
            assert torch.allclose(got[k], p.grad, rtol=1e-5, atol=1e-6), k


def test_shard_batch():
    from tamtr_amd.dist import shard_batch
    assert [shard_batch(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
