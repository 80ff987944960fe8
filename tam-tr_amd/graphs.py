"""HIP-graph replay of a static-shape part of the training step (forward and backward), MI355X.

Why not torch.cuda.make_graphed_callables: it differentiates with respect to the module's own parameters.  Their AccumulateGrad
nodes remember the stream they were created on; when the module has already run eagerly (default stream) and anything still
holds that graph (a stored loss, a gradient hook), the engine makes the *default* stream wait on the capturing stream while the
backward is being recorded, and hipStreamEndCapture segfaults (round 1's core dump; reproduced with tools/try_graph.py check).
Here the recorded function runs on detached aliases of the parameters (same storage, fresh leaves created on the capture
stream), warm-up and both captures use ONE side stream, and the real parameters only appear as inputs of the replaying
autograd node - nothing recorded ever touches an existing autograd graph or the default stream.

AQL packet capture (ROCm 7.2's default way of launching a graph: every kernel node's dispatch packet is built once and the replay
only rings the doorbell - hipGraphLaunch of the ~1 700-node forward graph costs the host 1 ms instead of 19, profiles/r04_host_phases.txt).
Rounds 2 - 3 ran with it switched OFF (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) because replays of the recorded backward returned garbage
gradients.  Round 4 found what breaks (profiles/r04_packet_capture_bisect.txt): MEMSET NODES.  A hipMemsetAsync recorded into a
graph is not kept in order with the kernel nodes around it once the packets are pre-built; every node that is a kernel replays exactly.
Two things put memset nodes into this recording: torch's multi-workgroup reductions (`partials.sum(0)`: Reduce.cuh zeroes its arrival
semaphores with cudaMemsetAsync before every launch and never resets them - the first replay finds fresh zeros, every later one a
stale count, and the output is never written) and MIOpen's weight-gradient / input-gradient solvers for 1x1 convolutions (memset +
atomic adds).  Neither the library GEMMs (rocBLAS and hipBLASLt replay correctly: the round-3 diagnosis was wrong) nor any kernel of
this package was involved.  The recorded part therefore contains no memset node any more - ordered slab sums (ops.slab_sum) instead of
torch reductions, the 1x1 convolutions' backward as GEMMs (ops._Conv1x1CL) - and packet capture stays ON.  GraphedPart checks this: it
counts the memset nodes of both graphs (hipGraphGetNodes on the capturing stream's graph) and refuses to build with any of them while packet capture is on, and
verify() replays three times (the corruption only shows from the second replay on).  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, exported before
the HIP runtime starts, restores node-by-node launches, under which memset nodes are harmless.
"""
import os

import torch


# torch.version.hip prefixes the replay path (GraphedPart.verify, packet capture on and off) was exercised on
VALIDATED_HIP = ('7.0', '7.2')


def packet_capture_on():
    """The HIP runtime launches graphs from pre-built AQL packets (its default) unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the
    environment when it started."""
    return os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '1') != '0'


def capture_census(stream):
    """Nodes by type of the graph `stream` is capturing into right now: {'kernel': n, 'memcpy': n, 'memset': n, 'other': n}
    (csrc/capi.hip tamtr_graph_capture_census).  Call inside the capture, after the last recorded op."""
    import ctypes
    from . import _lib
    counts = (ctypes.c_int * 16)()
    _lib.call('tamtr_graph_capture_census', ctypes.c_void_p(stream.cuda_stream), ctypes.cast(counts, ctypes.c_void_p), 16)
    c = list(counts)
    return {'kernel': c[0], 'memcpy': c[1], 'memset': c[2], 'other': sum(c[3:])}


class GraphedPart:
    """fn = GraphedPart(module, sample_args);  out = fn(*args) replays the recorded forward, out.backward() the recorded backward.

    module: nn.Module whose forward(*tensors) -> one tensor has shapes, dtypes and control flow fixed by the argument shapes.
    sample_args: CUDA tensors of those shapes (not differentiated).  Buffers the module updates in place (BatchNorm statistics)
    are updated by every replay, as in eager mode.  Gradients reach the module's parameters through ordinary AccumulateGrad."""

    def __init__(self, module, sample_args, warmup=3, log=None):
        if not all(isinstance(a, torch.Tensor) and a.is_cuda for a in sample_args):
            raise ValueError('sample_args must be CUDA tensors')
        self.module = module
        self.names, self.params = zip(*[(n, p) for n, p in module.named_parameters()])
        alias = {n: torch.nn.Parameter(p.detach(), requires_grad=p.requires_grad) for n, p in zip(self.names, self.params)}  # same storage
        # bf16 copies of the weights kept by the optimizer kernel (engine.FusedOptimStep(shadows=True)): the recorded function reads them through
        # the aliases (ops.bf16_shadow follows `_tamtr_alias_of` to the real parameter for the version check).  A replay runs no Python inside,
        # so __call__ re-derives any copy whose master was written behind the stepper's back BEFORE the replay (self.shadow_pairs).
        self.shadow_pairs = []
        for n, p in zip(self.names, self.params):
            sh = getattr(p, '_tamtr_bf16', None)
            if sh is not None and getattr(sh, '_tamtr_version', None) == p._version:
                alias[n]._tamtr_bf16, alias[n]._tamtr_alias_of = sh, p
                self.shadow_pairs.append((p, sh))
        self.static_in = [a.detach().clone() for a in sample_args]
        self.stream = torch.cuda.Stream()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()

        def run():
            return torch.func.functional_call(module, alias, tuple(self.static_in))

        import time
        t0 = time.perf_counter()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):   # MIOpen / hipBLASLt solver selection, lazy workspaces, the leaves' accumulators: all before capture
                out = run()
                used = [(n, a) for n, a in alias.items() if a.requires_grad]
                g = torch.autograd.grad(out, [a for _, a in used], torch.ones_like(out), allow_unused=True)
                del out, g
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.graph(self.fwd, pool=pool, stream=self.stream):
            self.static_out = run()
            census_f = capture_census(self.stream)
        self.static_gout = torch.zeros_like(self.static_out)
        leaves = [alias[n] for n in self.names if alias[n].requires_grad]
        with torch.cuda.graph(self.bwd, pool=pool, stream=self.stream):
            grads = torch.autograd.grad(self.static_out, leaves, self.static_gout, allow_unused=True)
            census_b = capture_census(self.stream)
        self.census = {'forward': census_f, 'backward': census_b}
        self.memset_nodes = (census_f['memset'], census_b['memset'])
        self.packet_capture = packet_capture_on()
        if self.packet_capture and any(self.memset_nodes):
            raise RuntimeError(f'GraphedPart: the recorded graphs contain memset nodes (forward {self.memset_nodes[0]}, backward {self.memset_nodes[1]}), which '
                               'do not replay in order under the HIP runtime\'s AQL packet capture (see the module docstring): something in the recorded '
                               'function zero-fills with hipMemsetAsync - torch.zeros / zero_(), a multi-workgroup torch reduction, a library solver.  '
                               'Remove it, or export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the HIP runtime starts')
        if log is not None:
            log(f'graph capture: {warmup} warm-up passes (kernel selection / compilation) {t1 - t0:.1f} s, recording forward + backward {time.perf_counter() - t1:.1f} s')
        it = iter(grads)
        self.static_grads = [next(it) if alias[n].requires_grad else None for n in self.names]
        self.n_live = sum(g is not None for g in self.static_grads)
        self._probe = next(((p, g) for p, g in zip(self.params, self.static_grads) if g is not None), None)
        part = self
        self.n_replays = 0   # forward replays made through __call__ (tests check that a step really took the recorded path)

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *flat):
                for dst, src in zip(part.static_in, flat[:len(part.static_in)]):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                part.fwd.replay()
                part.n_replays += 1
                return part.static_out.detach()

            @staticmethod
            @torch.autograd.function.once_differentiable
            def backward(ctx, gout):
                if gout.data_ptr() != part.static_gout.data_ptr():
                    part.static_gout.copy_(gout)
                part.bwd.replay()
                # the caller's AccumulateGrad adopts or adds these; they are rewritten by the next replay, i.e. after the optimizer
                # step that consumes them (one forward, one backward per step)
                return (None,) * len(part.static_in) + tuple(g.detach() if g is not None else None for g in part.static_grads)

        self._fn = _Replay

    @torch.no_grad()
    def _snapshot_buffers(self):
        return [b.detach().clone() for b in self.module.buffers()]

    @torch.no_grad()
    def _restore_buffers(self, saved):
        for b, v in zip(self.module.buffers(), saved):
            b.copy_(v)

    def verify(self, replays=3, tol=2e-2, noise_factor=6.0, junk_between=True):
        """Replay the two recorded graphs on the capture inputs and hold them to an EAGER forward + backward of the same module on the
        same inputs and the same cotangent: the output and every live parameter gradient.  The eager pass runs twice, so every figure
        comes with the eager run-to-run level of the same quantity next to it: with MIOpen's split-K / atomic solvers the trunk is not
        bitwise reproducible and its BatchNorm stack amplifies that - gradients that are sums with heavy cancellation (biases) then
        differ by O(1) of their own size between two EAGER runs and carry no information; on deterministic solvers
        (tuning.use_deterministic_convolutions) eager is reproducible and a replay equals it bit for bit
        (tests/test_gpu_graphs.py).  Metrics are relative L2 norms: per tensor (`*_rel`) and over all gradients together (`grad_l2_rel`).
        `ok` is False when anything is non-finite, when the output or the whole gradient is off by more than
        max(tol, noise_factor * eager level), or when an INFORMATIVE tensor (eager level < 0.1) is off by more than that bound on its
        own level.  When eager itself is not reproducible to 5 % (`conclusive: False`) only "finite and within 10 x" is decided.  A replay that went wrong is NaN or off by many orders of magnitude (profiles/r03_graph_probe.txt).
        Buffers the module updates in place (BatchNorm statistics) are restored afterwards; no global RNG is consumed (the module
        must be a function of its inputs: DropPath factors are inputs)."""
        saved = self._snapshot_buffers()
        live = [i for i, g in enumerate(self.static_grads) if g is not None]
        gen = torch.Generator(device=self.static_out.device).manual_seed(20261004)
        cot = torch.randn(self.static_out.shape, generator=gen, device=self.static_out.device, dtype=torch.float32).to(self.static_out.dtype)

        def eager():
            self._restore_buffers(saved)
            with torch.enable_grad():
                out = self.module(*self.static_in)
                gr = torch.autograd.grad(out, [self.params[i] for i in live], cot, allow_unused=True)
            return out.detach().float().clone(), [None if g is None else g.detach().float().clone() for g in gr]

        def dist(a, b):   # (|a - b|^2, |b|^2) as python floats; nan-safe
            if a is None or b is None:
                return (0.0, 0.0) if a is b else (float('inf'), 1.0)
            a, b = a.detach(), b.detach()
            d = (a.float() - b.float()).double()
            return float((d * d).sum()), float((b.double() * b.double()).sum())

        def rel(num, den):
            return (num / den) ** 0.5 if den > 0 else (0.0 if num == 0 else float('inf'))

        o1, g1 = eager()
        o2, g2 = eager()
        noise_out = rel(*dist(o2, o1))
        nz = [dist(a, b) for a, b in zip(g2, g1)]
        noise_t = [rel(*x) for x in nz]
        noise_all = rel(sum(x[0] for x in nz), sum(x[1] for x in nz))
        informative = [n == n and n < 0.1 for n in noise_t]
        res = {'grads': len(live), 'informative_grads': sum(informative), 'eager_noise_out': noise_out, 'eager_noise_grad_l2': noise_all,
               'eager_noise_grad_max': max(noise_t) if noise_t else 0.0, 'replays': []}
        ok = True
        # Two regimes.  Eager reproducible to a few per cent (the tuned tables: 2e-3; deterministic solvers: 0): the bound is tight and the
        # check is conclusive.  Eager itself all over the place (MIOpen's heuristic solvers on this graph: two eager runs differ by 0.1 - 1
        # in the whole-gradient norm): nothing finer than "finite and not orders of magnitude off" can be decided - the result says so
        # (`conclusive: False`); a replay gone wrong in the way round 2 saw it (1e20 .. NaN) is still caught.
        conclusive = noise_all < 0.05 and noise_out < 0.05
        if conclusive:
            b_out, b_all = max(tol, noise_factor * noise_out), max(tol, noise_factor * noise_all)
        else:
            b_out = b_all = 10.0
        for rep in range(replays):
            self._restore_buffers(saved)
            self.static_gout.copy_(cot)
            self.fwd.replay()
            if junk_between:  # eager allocations and kernels between the two replays, as the decoder and the loss make them in a step
                junk = torch.full((1 << 22,), float('nan'), device=cot.device)
                del junk
            self.bwd.replay()
            e_out = rel(*dist(self.static_out, o1))
            ds = [dist(self.static_grads[i], g) for i, g in zip(live, g1)]
            errs = [rel(*x) for x in ds]
            e_all = rel(sum(x[0] for x in ds), sum(x[1] for x in ds))
            nonfinite = sum(1 for i in live if not bool(torch.isfinite(self.static_grads[i]).all()))
            over = [(errs[j] / max(tol, noise_factor * noise_t[j]), j) for j in range(len(errs)) if informative[j]]
            over = [(v if v == v else float('inf'), j) for v, j in over]
            worst = max(over) if over else (0.0, None)
            rec = {'out_rel': e_out, 'grad_l2_rel': e_all, 'grad_rel_max': max(errs) if errs else 0.0,
                   'worst_informative_grad': None if worst[1] is None else self.names[live[worst[1]]],
                   'worst_informative_over_bound': worst[0], 'nonfinite_grads': nonfinite}
            res['replays'].append(rec)
            bad = nonfinite or not (e_out <= b_out) or not (e_all <= b_all) or not (worst[0] <= 1.0)
            ok = ok and not bad
        self._restore_buffers(saved)
        res.update(ok=ok, conclusive=conclusive, bound_out=b_out, bound_grad_l2=b_all, out_rel_max=max(r['out_rel'] for r in res['replays']),
                   grad_l2_rel_max=max(r['grad_l2_rel'] for r in res['replays']), grad_rel_max=max(r['grad_rel_max'] for r in res['replays']))
        return res

    def __call__(self, *args):
        if self._probe is not None and self._probe[0].grad is not None and self._probe[0].grad.data_ptr() == self._probe[1].data_ptr():
            # AccumulateGrad adopted the static gradient buffer as .grad last step; accumulating onto it would add the new
            # gradient to itself
            raise RuntimeError('GraphedPart: clear gradients with zero_grad(set_to_none=True) (or give .grad its own storage) between steps')
        if self.shadow_pairs:
            stale = [(p, s) for p, s in self.shadow_pairs if s._tamtr_version != p._version]
            if stale:   # (not on the normal path: FusedOptimStep.step() keeps the copies current)
                with torch.no_grad():
                    torch._foreach_copy_([s for _, s in stale], [p for p, _ in stale])
                for p_, s_ in stale:
                    s_._tamtr_version = p_._version
        return self._fn.apply(*args, *self.params)
