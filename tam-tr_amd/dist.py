"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Reference semantics (ultralytics/engine/trainer.py:200-204,241,252,346-347): images are sharded across ranks, every rank
runs the same graph, gradients are averaged by DDP and the loss is pre-multiplied by world_size, i.e. the update uses the
SUM over ranks of the per-rank gradients.  Here: an explicit bucketed gradient reducer instead of the DDP wrapper -
  * parameters that never receive a gradient (the 30 `model.{16,24,32,36,40}.attn.*` tensors) are excluded up
    front, which is what makes plain DDP fail on iteration 2 in the reference;
  * gradients live in a few large flat buckets (views), one all-reduce per bucket, launched from a post-accumulate hook
    once the bucket's last gradient of this backward pass is written AND every earlier bucket has been launched, so the
    collectives overlap the rest of the backward and every rank issues them in the same (bucket index) order whatever
    its data did - a rank whose batch has no GT boxes never produces a gradient for `denoising_class_embed`, and
    collectives are matched by issue order; xGMI is point-to-point (7 links per GPU), so few large buckets (default
    32 MiB) beat many small ones;
  * no collective anywhere on the data path of the forward (per-rank BatchNorm statistics, like the reference).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, solo=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world).
    solo: create the process group even for ONE rank (a one-rank RCCL communicator: every collective of the N > 1 path is
    issued through the real backend - what a one-GPU box can show of it; two ranks cannot share a device under RCCL)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or solo) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1]) if world == 1 else '29500'
        if backend is None:
            backend = os.environ.get('TAMTR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def launch_plan(gpus, environ, argv, script):
    """The command with which a script started as plain `python script.py --gpus N ...` becomes the launcher of its own N ranks - what
    the reference's trainer does for device='0,1,..' (engine/trainer.py:161-189 -> utils/dist.py:50-62: subprocess.run of
    `python -m torch.distributed.run --nproc_per_node N ... file`).  None when there is nothing to start: one GPU, or RANK / WORLD_SIZE
    already in the environment (some launcher did it).  A pure function of its arguments (tests/test_host_logic.py) apart from the
    free rendezvous port it picks on 127.0.0.1."""
    import socket
    import sys
    if gpus <= 1 or 'WORLD_SIZE' in environ or 'RANK' in environ:
        return None
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(script), *argv]


def self_launch(cmd, environ=None):
    """Run the ranks as a CHILD process (never an exec of this one), with stdout / stderr passed through; returns their exit code.
    Must be called before this process touches the GPU: it only waits."""
    import subprocess
    import sys
    print(f'[launch] {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=dict(os.environ if environ is None else environ)).returncode


def shard_batch(global_batch, rank, world):
    """Contiguous image shard [lo, hi) of rank (DistributedSampler-style equal split; global_batch % world == 0)."""
    if global_batch % world:
        raise ValueError(f'global batch {global_batch} not divisible by world size {world}')
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradReducer:
    """Bucketed all-reduce (SUM or MEAN) of the gradients of `params` that are known to receive one.

    Each bucket is one flat buffer.  Autograd writes every gradient wherever it likes (for the graph-replayed part of the step these
    are the graph's own output buffers, adopted by AccumulateGrad without a copy); when the last gradient of a bucket has arrived -
    and every earlier bucket has been launched - the bucket is gathered with ONE multi-tensor copy, all-reduced asynchronously, and
    the parameters' .grad are re-pointed at their slices of the flat buffer, so the optimizer reads reduced values without a copy
    back.  (Round 1 pre-set .grad to the slices and let autograd accumulate into them: one add kernel per parameter per step plus a
    zero fill of every bucket - ~700 launches.)"""

    def __init__(self, named_params, bucket_bytes=32 << 20, op='sum', grad_dtype=None, skip=lambda name: False,
                 late=lambda name: False, always_collective=False):
        """grad_dtype: dtype of the buckets on the wire (torch.bfloat16 halves the 168.5 MB of fp32 gradients per step;
        the optimizer still sees fp32 gradients).  skip(name): parameters that never get a gradient.  late(name):
        parameters that may get none on some steps (`denoising_class_embed` when a rank's batch has no boxes) - they go
        into the last bucket, so that a missing hook delays only that bucket's launch to finish().  always_collective: issue the
        all-reduces on a one-rank group too (the RCCL rehearsal on one GPU: same calls, same streams, nothing to exchange)."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        self.op = op
        self.buckets = []   # dicts: flat, params, views, pending, handle
        self._hooks = []
        self._next = 0      # index of the first bucket not yet handed to the collective (launch order == bucket order)
        params = [(n, p) for n, p in named_params if p.requires_grad and not skip(n)]
        params.reverse()  # roughly the order gradients become ready in backward
        params = [x for x in params if not late(x[0])] + [x for x in params if late(x[0])]
        cur, cur_bytes = [], 0
        groups = []
        for n, p in params:
            nbytes = p.numel() * (torch.tensor([], dtype=grad_dtype or p.dtype).element_size())
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append((n, p))
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for g in groups:
            dt = grad_dtype or g[0][1].dtype
            flat = torch.zeros(sum(p.numel() for _, p in g), device=g[0][1].device, dtype=dt)
            off = 0
            bucket = {'flat': flat, 'params': g, 'pending': 0, 'handle': None, 'views': [], 'wide': None, 'wide_views': []}
            if any(p.dtype != dt for _, p in g):
                # a narrower wire dtype: the reduced bucket is widened back with ONE copy into a twin buffer of the parameters' dtype, whose
                # slices become .grad (round 4: was one cast launch per parameter, ~600 per step, 0.8 ms on the one-rank RCCL rehearsal)
                if len({p.dtype for _, p in g}) != 1:
                    raise ValueError('a gradient bucket with a wire dtype needs parameters of one dtype')
                bucket['wide'] = torch.zeros_like(flat, dtype=g[0][1].dtype)
            for _, p in g:
                # same strides as the parameter (channels-last conv weights): the optimizer stays on its fast path
                for buf, key in ((flat, 'views'), (bucket['wide'], 'wide_views')):
                    if buf is not None:
                        bucket[key].append(buf[off:off + p.numel()].view_as(p) if p.is_contiguous() else torch.as_strided(buf, p.shape, p.stride(), off))
                off += p.numel()
            self.buckets.append(bucket)
            for _, p in g:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bucket)))
        self.n_params = sum(len(b['params']) for b in self.buckets)

    def _make_hook(self, bucket):
        def hook(param):
            bucket['pending'] -= 1
            if bucket['pending'] == 0:
                self._launch_ready()
        return hook

    def _gather(self, b):
        """This backward's gradients of the bucket's parameters -> the flat buffer (one multi-tensor copy, casting if the wire
        dtype differs); a parameter without a gradient on this rank contributes zeros."""
        dst, src = [], []
        for (_, p), v in zip(b['params'], b['views']):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        for (_, p), v in zip(b['params'], b['views']):
            p.grad = v if v.dtype == p.dtype else None   # finish() hands back the reduced gradient of a narrower wire dtype

    def _launch_ready(self):
        """Launch, in index order, every complete bucket that directly follows the launched prefix.  A complete bucket
        behind an incomplete one waits (for that one's last hook, or for finish()): the issue order is rank-independent."""
        while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
            b = self.buckets[self._next]
            self._gather(b)
            if self.collective:
                b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, async_op=True)
            self._next += 1

    def prepare(self):
        """Call before backward: arm the per-bucket counters and detach .grad from the flat buffers (autograd must not accumulate
        onto last step's reduced values; fresh gradients are gathered when their bucket completes)."""
        self._next = 0
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['handle'] = None
            for _, p in b['params']:
                p.grad = None

    def finish(self):
        """Call after backward: launch what is left in index order (a bucket with a parameter that got no gradient on this
        rank holds zeros for it, and holds back the buckets behind it until here), then wait for all of them."""
        for b in self.buckets[self._next:]:
            b['pending'] = 0
        self._launch_ready()
        for b in self.buckets:
            if self.collective:
                b['handle'].wait()
                if self.op == 'mean' and self.world > 1:
                    b['flat'].div_(self.world)
            if b['wide'] is not None:
                b['wide'].copy_(b['flat'])
                for (_, p), v in zip(b['params'], b['wide_views']):
                    p.grad = v

    def remove(self):
        for h in self._hooks:
            h.remove()
