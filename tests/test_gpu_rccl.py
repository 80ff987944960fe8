"""MI355X, ONE rank on the RCCL backend (`nccl`): what a one-GPU box can show of the N > 1 path on the real backend (two ranks cannot
share a device under RCCL; the two- and eight-rank cases run over gloo in tests/test_dist_gloo.py).  The gradient reducer is forced to
issue its collectives on the one-rank communicator: bucket gather -> ncclAllReduce on RCCL's stream -> wait -> the optimizer's view, from
the post-accumulate hooks (autograd thread), with fp32 and bf16 wire buckets, incl. a parameter without a gradient and a channels-last
weight.  A one-rank SUM is the identity, so the reduced gradients must equal the plain ones bit for bit (bf16 wire: after one rounding)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu


class Small(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(8, 16, 3, padding=1)
        self.a = nn.Linear(16, 32)
        self.unused = nn.Linear(8, 8)      # never contributes (the 30 discarded-gate parameters): skipped
        self.late = nn.Embedding(4, 32)    # no gradient on this step (`denoising_class_embed` on a batch without boxes)
        self.b = nn.Linear(32, 4)

    def forward(self, x):
        y = self.conv(x).mean((2, 3))
        return self.b(torch.relu(self.a(y)))


@pytest.fixture(scope='module')
def rccl_solo():
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('wire', ['fp32', 'bf16'])
def test_reducer_issues_its_buckets_on_a_one_rank_rccl_group(rccl_solo, wire):
    from tamtr_amd import dist as tdist
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    torch.manual_seed(0)
    m = Small().cuda().to(memory_format=torch.channels_last)
    x = torch.randn(4, 8, 12, 12, device='cuda')
    # the gradients as autograd hands them over in the SAME backward pass (the convolution's weight gradient is not bitwise repeatable
    # from one pass to the next: atomic split sums), taken by tensor hooks in front of the reducer's post-accumulate hooks
    plain = {}
    taps = [p.register_hook(lambda g, n=n: plain.__setitem__(n, g.clone())) for n, p in m.named_parameters()]

    calls = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        calls.append((t.dtype, t.numel()))
        return real(t, *a, **k)

    red = tdist.GradReducer(m.named_parameters(), bucket_bytes=2048, skip=lambda n: n.startswith('unused'), late=lambda n: n.startswith('late'),
                            grad_dtype=torch.bfloat16 if wire == 'bf16' else None, always_collective=True)
    assert red.collective and len(red.buckets) >= 3
    dist.all_reduce = spy
    try:
        for _ in range(2):      # second step: .grad re-pointed at the flat buffers by the first one
            calls.clear()
            plain.clear()
            red.prepare()
            m(x).square().sum().backward()
            red.finish()
            torch.cuda.synchronize()
            assert len(calls) == len(red.buckets), 'one collective per bucket'
            assert 'late.weight' not in plain and 'unused.weight' not in plain and len(plain) == 6
            assert all(dt == (torch.bfloat16 if wire == 'bf16' else torch.float32) for dt, _ in calls)
            for n, p in m.named_parameters():
                if n.startswith('unused'):
                    assert p.grad is None
                elif n.startswith('late'):
                    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0     # zeros travel for it
                else:
                    want = plain[n] if wire == 'fp32' else plain[n].bfloat16().float()
                    assert p.grad.dtype == torch.float32 and p.grad.stride() == p.stride()
                    assert torch.equal(p.grad, want), n
    finally:
        dist.all_reduce = real
        red.remove()
        for h in taps:
            h.remove()


def test_collectives_of_the_bench_path_on_one_rank_rccl(rccl_solo):
    """barrier, MIN agreement on the capture result, MAX over ranks of the step time: the calls bench.py makes around the timed region."""
    dist.barrier()
    ok = torch.tensor([1], device='cuda', dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    t = torch.tensor([0.125], device='cuda', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    assert int(ok) == 1 and float(t) == 0.125


def test_bench_stdout_is_one_json_line_with_an_rccl_group():
    """RCCL prints a version banner to stdout when the communicator is created; bench.py's stdout must still be exactly the JSON line the
    driver parses (file descriptor 1 is pointed at stderr for the run, the line goes to the saved descriptor).  The bench at its own
    shapes, two timed steps, kernel-by-kernel static part (no capture: seconds), as a child process with a one-rank RCCL group."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--rccl-solo', '--no-cpu-baseline', '--no-graph-check', '--static-part', 'eager',
                        '--steps', '2', '--warmup', '1'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines[:8]
    d = json.loads(lines[0])
    assert d['config']['dist_backend'] == 'nccl' and d['n_gpus'] == 1 and 'rccl_solo_rehearsal' in d['config'] and d['ms_per_step'] > 0
    assert 'RCCL version' in r.stderr or 'RCCL' not in r.stdout      # the banner, if this RCCL build prints one, went to stderr
