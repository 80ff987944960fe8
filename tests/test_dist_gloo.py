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


class Gappy(nn.Module):
    """`maybe` (first in parameter order = last, hence 'bucket-closing', in the reducer's order) contributes only when the
    batch says so - the shape of `denoising_class_embed` on a rank whose batch has no GT boxes (loss.get_cdn_group -> None)."""

    def __init__(self):
        super().__init__()
        self.maybe = nn.Linear(8, 8)
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.c = nn.Linear(4, 4)

    def forward(self, x, use_maybe):
        y = self.c(self.b(torch.relu(self.a(x))))
        return y + self.maybe(x)[:, :4] if use_maybe else y


def _gap_worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tamtr_amd.dist import GradReducer, init_from_env, shard_batch
    init_from_env('gloo')
    torch.manual_seed(0)
    model = Gappy()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(1))
    lo, hi = shard_batch(8, rank, world)
    if mode == 'gap_first':     # the grad-less parameter closes bucket 0 on rank 1 only: launch orders used to diverge here
        red = GradReducer([(n, p) for n, p in model.named_parameters()][::-1], bucket_bytes=300)
        assert len(red.buckets) >= 3 and red.buckets[0]['params'][0][0].startswith('maybe')
    elif mode == 'late':        # same, with the parameter declared `late`: it sits in the last bucket
        red = GradReducer([(n, p) for n, p in model.named_parameters()][::-1], bucket_bytes=300, late=lambda n: n.startswith('maybe'))
        assert all(n.startswith('maybe') for n, _ in red.buckets[-1]['params'][-2:])
    else:                       # bf16 buckets on the wire, fp32 gradients for the optimizer
        red = GradReducer(model.named_parameters(), bucket_bytes=300, grad_dtype=torch.bfloat16)
        assert all(b['flat'].dtype == torch.bfloat16 for b in red.buckets)
    order = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        order.append(t.numel())
        return real(t, *a, **k)
    dist.all_reduce = spy
    grads = []
    for step in range(3):
        red.prepare()
        use = not (rank == 1 and step == 1)          # step 1: rank 1 has "no boxes"
        model(x[lo:hi], use).pow(2).sum().backward()
        if mode == 'late' and step == 1 and rank == 1:
            assert len(order) % len(red.buckets) == len(red.buckets) - 1      # everything but the last bucket already went out
        red.finish()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
        assert all(g.dtype == torch.float32 for g in grads[-1].values())
    dist.all_reduce = real
    assert order == [b['flat'].numel() for b in red.buckets] * 3, order   # bucket index order on every rank, every step
    if rank == 0:
        torch.save(grads, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['gap_first', 'late', 'bf16'])
def test_two_rank_reducer_with_a_gradless_parameter_on_one_rank(tmp_path, mode):
    """The collectives go out in bucket-index order on every rank even when a rank's step leaves a bucket parameter without
    a gradient (NCCL/gloo match collectives by issue order), and the sums are those of the single process."""
    out = str(tmp_path / 'g.pt')
    mp.spawn(_gap_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Gappy()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(1))
    for step in range(3):
        model.zero_grad()
        (model(x[:4], True).pow(2).sum() + model(x[4:], step != 1).pow(2).sum()).backward()
        for k, p in model.named_parameters():
            tol = dict(rtol=2e-2, atol=2e-2) if mode == 'bf16' else dict(rtol=1e-5, atol=1e-6)   # bf16: 8 significant bits per addend
            assert torch.allclose(got[step][k], p.grad, **tol), (step, k)
            if mode == 'bf16':      # and it IS the bf16-rounded sum, not something looser
                assert not torch.equal(got[step][k], p.grad) or float(p.grad.abs().max()) == 0


def test_shard_batch():
    from tamtr_amd.dist import shard_batch
    assert [shard_batch(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def _fit_worker(rank, world, port, img_dir, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from test_engine import _ToyDetector
    from tamtr_amd import data as D
    from tamtr_amd.dist import GradReducer, init_from_env
    from tamtr_amd.engine import fit
    init_from_env('gloo')
    names = ['a', 'b', 'c', 'd', 'e']
    ds = D.PromptDetDataset(img_dir, names, imgsz=32, augment=True, batch_size=2)
    loader = D.build_dataloader(ds, 2, workers=0, shuffle=True, rank=rank)
    val = D.build_dataloader(D.PromptDetDataset(img_dir, names, imgsz=32, augment=False), 4, workers=0, shuffle=False) if rank == 0 else None
    tf = D.TextFeatures.synthetic(names + [''], dim=8)
    torch.manual_seed(0)
    model = _ToyDetector()
    start = {k: v.clone() for k, v in model.state_dict().items()}
    seen = []

    def prepare(batch, training):
        if training:
            seen.extend(batch['im_file'])
        return D.preprocess_batch(batch, tf if training else None, 'cpu')
    red = GradReducer(model.named_parameters(), bucket_bytes=1 << 10)
    hist = fit(model, loader, prepare, epochs=2, val_loader=val, lr0=1e-2, warmup_iters=0, imgsz=32, reducer=red, rank=rank, world=world,
               save_dir=os.path.join(out, 'run') if rank == 0 else None)
    torch.save({'state': model.state_dict(), 'start': start, 'seen': seen, 'hist': hist}, os.path.join(out, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fit_shards_images_and_keeps_ranks_in_step(tmp_path):
    """engine.fit over data.build_dataloader(rank=...) with the gradient reducer: every epoch each rank trains on its own half of
    the images, the weights stay identical across ranks, only rank 0 validates and writes checkpoints."""
    import numpy as np
    from PIL import Image
    g = np.random.default_rng(3)
    (tmp_path / 'images').mkdir(), (tmp_path / 'labels').mkdir()
    for i in range(8):
        Image.fromarray(g.integers(0, 255, (40, 48, 3), dtype=np.uint8)).save(tmp_path / 'images' / f'{i}.png')
        (tmp_path / 'labels' / f'{i}.txt').write_text(f'{i % 5} 0.5 0.5 0.3 0.3\n')
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path / 'images'), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    for k, v in r0['state'].items():
        if 'running_' in k or 'num_batches' in k:
            continue                                    # BatchNorm statistics are per rank (no SyncBN), as in the reference
        assert torch.equal(v, r1['state'][k]), k
    assert any(not torch.equal(r0['state'][k], r0['start'][k]) for k in ('head.weight', 'conv.weight'))
    for e in range(2):                                  # 8 images, 2 ranks, batch 2 -> 2 steps per rank and epoch
        a, b = set(r0['seen'][4 * e:4 * e + 4]), set(r1['seen'][4 * e:4 * e + 4])
        assert len(a) == 4 and len(b) == 4 and not (a & b)
    assert r0['hist'][-1]['steps'] == 4 and 'mAP50' in r0['hist'][-1] and 'mAP50' not in r1['hist'][-1]
    assert (tmp_path / 'run' / 'last.pt').exists()


def test_reducer_buckets_keep_channels_last_strides_single_process():
    """World size 1 (no process group): the reducer still gathers gradients into its flat buckets; the slice of a channels-last conv
    weight (NHWC trunk) carries the parameter's strides, so .grad and the parameter stay layout-compatible for the fused optimizer,
    and the values equal plain autograd's."""
    from tamtr_amd.dist import GradReducer
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(4, 8, 3, padding=1)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(2, 4, 6, 6)
    conv(x).pow(2).sum().backward()
    want = {k: p.grad.clone() for k, p in conv.named_parameters()}
    conv.zero_grad(set_to_none=True)
    red = GradReducer(conv.named_parameters(), bucket_bytes=64)
    for _ in range(2):
        red.prepare()
        assert conv.weight.grad is None
        conv(x).pow(2).sum().backward()
        red.finish()
        assert conv.weight.grad.stride() == conv.weight.stride() and not conv.weight.grad.is_contiguous()
        for k, p in conv.named_parameters():
            assert torch.allclose(p.grad, want[k], rtol=1e-6, atol=1e-6), k
            flat = [b['flat'] for b in red.buckets if any(q is p for _, q in b['params'])][0]
            assert flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * flat.element_size()
    red.remove()


def test_one_rank_group_issues_every_collective_when_asked(monkeypatch):
    """`always_collective` (bench.py --rccl-solo: the one-rank rehearsal on the real backend) - here on a one-rank gloo group: one
    all-reduce per bucket, values untouched; without the flag a one-rank group issues none.  `init_from_env(solo=True)` makes the group."""
    from tamtr_amd import dist as tdist
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        monkeypatch.delenv(k, raising=False)
    assert tdist.init_from_env(backend='gloo', solo=True) == (0, 0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    try:
        torch.manual_seed(0)
        net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
        x = torch.randn(5, 8)
        net(x).sum().backward()
        want = [p.grad.clone() for p in net.parameters()]
        calls = []
        real = dist.all_reduce
        monkeypatch.setattr(dist, 'all_reduce', lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1])
        for flag in (False, True):
            red = tdist.GradReducer(net.named_parameters(), bucket_bytes=128, always_collective=flag)
            calls.clear()
            red.prepare()
            net(x).sum().backward()
            red.finish()
            assert len(calls) == (len(red.buckets) if flag else 0) and len(red.buckets) > 1
            for p, w in zip(net.parameters(), want):
                assert torch.equal(p.grad, w)
            red.remove()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ `--gpus N` starts its own ranks
def test_launch_plan_decision_and_argv():
    """bench.py / tools/train.py started as plain `python script --gpus N` become the launcher of N ranks (reference:
    engine/trainer.py:161-189 -> utils/dist.py:50-62); under a launcher (RANK / WORLD_SIZE set) or for one GPU they do not."""
    from tamtr_amd.dist import launch_plan
    script = os.path.join(ROOT, 'bench.py')
    assert launch_plan(1, {}, ['--gpus', '1'], script) is None
    assert launch_plan(4, {'WORLD_SIZE': '4', 'RANK': '0'}, ['--gpus', '4'], script) is None          # the driver's torch.distributed.run form
    assert launch_plan(4, {'RANK': '2'}, ['--gpus', '4'], script) is None
    cmd = launch_plan(4, {'PATH': '/usr/bin'}, ['--gpus', '4', '--steps', '20', '--warmup', '5'], 'bench.py')
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    port = int(cmd[cmd.index('--master-port') + 1])
    assert 1024 < port < 65536
    k = cmd.index(os.path.abspath('bench.py'))                 # the script by absolute path, then the caller's own arguments unchanged
    assert cmd[k + 1:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']


def test_self_launch_runs_two_gloo_ranks_and_returns_their_code(tmp_path):
    """The launch itself, on the CPU: a script that calls launch_plan / self_launch the way bench.py does comes back as two gloo ranks
    that see each other; a failing rank's exit code is propagated."""
    script = tmp_path / 'mini.py'
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
from tamtr_amd import dist as tdist
plan = tdist.launch_plan(int(sys.argv[sys.argv.index('--gpus') + 1]), os.environ, sys.argv[1:], __file__)
if plan is not None:
    raise SystemExit(tdist.self_launch(plan))
import torch, torch.distributed as dist
rank, local, world = tdist.init_from_env('gloo')
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == 0:
    open(sys.argv[sys.argv.index('--out') + 1], 'w').write(f'{{world}} {{float(t)}}')
dist.barrier()
dist.destroy_process_group()
raise SystemExit(3 if '--fail' in sys.argv else 0)
''')
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = tmp_path / 'o.txt'
    r = subprocess.run([sys.executable, str(script), '--gpus', '2', '--out', str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == '2 3.0'
    r = subprocess.run([sys.executable, str(script), '--gpus', '2', '--out', str(out), '--fail'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_eight_ranks_self_launched_bf16_buckets_one_rank_without_boxes(tmp_path):
    """The 8-GPU configuration (BASELINE configs[3]) rehearsed on the CPU: a script started as `python script --gpus 8` becomes the
    launcher of 8 gloo ranks (launch_plan / self_launch, as bench.py does); the ranks shard 64 samples, reduce their gradients through
    GradReducer with bf16 buckets on the wire, one rank's step has no contribution for the `late` parameter (a batch without boxes:
    no gradient for `denoising_class_embed`), every rank has a MIOpen table directory of its own (tuning.use_tuned_convolutions_ranked).  Checked: every rank issues the collectives in bucket order on every step, every rank ends with the same fp32
    gradients, and they are the bf16-wire sums of the single-process gradient of the whole batch."""
    script = tmp_path / 'dp8.py'
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
from tamtr_amd import dist as tdist
plan = tdist.launch_plan(int(sys.argv[sys.argv.index('--gpus') + 1]), os.environ, sys.argv[1:], __file__)
if plan is not None:
    raise SystemExit(tdist.self_launch(plan))
import torch, torch.nn as nn, torch.distributed as dist
rank, local, world = tdist.init_from_env('gloo')
torch.set_num_threads(1)
out = sys.argv[sys.argv.index('--out') + 1]
# every rank works on a MIOpen table directory of its own (tuning.use_tuned_convolutions_ranked: ranks that share one directory contend for
# MIOpen's lock files, and a rank that loses the race searches instead of looking up)
os.environ['TAMTR_MIOPEN_DB_DIR'] = out
from tamtr_amd import tuning
mine, _ = tuning._user_db_dir(f'-rank{{rank}}')
open(os.path.join(mine, 'owner'), 'w').write(str(rank))
dist.barrier()
dirs = sorted(d for d in os.listdir(out) if '-rank' in d)
assert len(dirs) == world, dirs
assert open(os.path.join(mine, 'owner')).read() == str(rank)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(11, 8)          # the `late` parameter: no gradient on a rank whose batch has no boxes
        self.a, self.b = nn.Linear(8, 32), nn.Linear(32, 4)
        self.dead = nn.Linear(8, 8)                # never differentiated: the discarded gates' parameters (skip)

    def forward(self, x, labels):
        y = self.b(torch.relu(self.a(x)))
        with torch.no_grad():
            self.dead(x)
        return y if labels is None else y + self.embed(labels)[:, :4]


torch.manual_seed(0)
model = Net()
g = torch.Generator().manual_seed(1)
x, lab = torch.randn(64, 8, generator=g), torch.randint(0, 11, (64,), generator=g)
lo, hi = tdist.shard_batch(64, rank, world)
red = tdist.GradReducer(model.named_parameters(), bucket_bytes=600, grad_dtype=torch.bfloat16, skip=lambda n: n.startswith('dead'),
                        late=lambda n: n.startswith('embed'))
assert len(red.buckets) >= 2 and red.buckets[-1]['params'][-1][0] == 'embed.weight'
order, real = [], dist.all_reduce
def spy(t, *a, **k):
    order.append(t.numel())
    return real(t, *a, **k)
dist.all_reduce = spy
grads = []
for step in range(3):
    red.prepare()
    boxes = not (rank == 5 and step == 1)          # step 1: rank 5's images have no boxes
    model(x[lo:hi], lab[lo:hi] if boxes else None).pow(2).sum().backward()
    red.finish()
    grads.append({{k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}})
dist.all_reduce = real
assert order == [b['flat'].numel() for b in red.buckets] * 3, order
torch.save(grads, os.path.join(out, f'rank{{rank}}.pt'))
dist.barrier()
dist.destroy_process_group()
''')
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, str(script), '--gpus', '8', '--out', str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = [torch.load(tmp_path / f'rank{k}.pt') for k in range(8)]

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = nn.Embedding(11, 8)
            self.a, self.b = nn.Linear(8, 32), nn.Linear(32, 4)
            self.dead = nn.Linear(8, 8)

        def forward(self, x, labels):
            y = self.b(torch.relu(self.a(x)))
            return y if labels is None else y + self.embed(labels)[:, :4]
    torch.manual_seed(0)
    model = Net()
    g = torch.Generator().manual_seed(1)
    x, lab = torch.randn(64, 8, generator=g), torch.randint(0, 11, (64,), generator=g)
    for step in range(3):
        model.zero_grad()
        sum(model(x[8 * k:8 * k + 8], None if (k == 5 and step == 1) else lab[8 * k:8 * k + 8]).pow(2).sum() for k in range(8)).backward()
        for k in range(1, 8):       # all ranks hold the same reduced gradients, in fp32, for exactly the reduced parameters
            assert set(got[k][step]) == set(got[0][step]) == {'embed.weight', 'a.weight', 'a.bias', 'b.weight', 'b.bias'}
            assert all(torch.equal(got[k][step][n], got[0][step][n]) and got[k][step][n].dtype == torch.float32 for n in got[0][step])
        for n, p in model.named_parameters():
            if n in got[0][step]:   # 8 bf16 addends per element: 8 significant bits each
                assert torch.allclose(got[0][step][n], p.grad, rtol=3e-2, atol=3e-2 * float(p.grad.abs().max())), (step, n)
