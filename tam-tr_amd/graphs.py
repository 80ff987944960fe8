"""HIP-graph replay of a static-shape part of the training step (forward and backward), MI355X.

Why not torch.cuda.make_graphed_callables: it differentiates with respect to the module's own parameters.  Their AccumulateGrad
nodes remember the stream they were created on; when the module has already run eagerly (default stream) and anything still
holds that graph (a stored loss, a gradient hook), the engine makes the *default* stream wait on the capturing stream while the
backward is being recorded, and hipStreamEndCapture segfaults (round 1's core dump; reproduced with tools/try_graph.py check).
Here the recorded function runs on detached aliases of the parameters (same storage, fresh leaves created on the capture
stream), warm-up and both captures use ONE side stream, and the real parameters only appear as inputs of the replaying
autograd node - nothing recorded ever touches an existing autograd graph or the default stream.

Runtime switch: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (set by the package __init__ before HIP starts).  With ROCm 7.2's packet capture
of graph nodes on, replays of the recorded backward intermittently returned garbage / NaN gradients (VSS blocks + projection,
deterministic in eager mode: replay 4 of 5 wrong; 12 of 12 identical to eager with the switch off; a plain memset / reduce chain
replays correctly either way).  Without it hipGraphLaunch enqueues node by node (the host is busy ~100 ms per step inside the two
launches) but the GPU never waits: the step runs at its kernel time.
"""
import os

import torch


class GraphedPart:
    """fn = GraphedPart(module, sample_args);  out = fn(*args) replays the recorded forward, out.backward() the recorded backward.

    module: nn.Module whose forward(*tensors) -> one tensor has shapes, dtypes and control flow fixed by the argument shapes.
    sample_args: CUDA tensors of those shapes (not differentiated).  Buffers the module updates in place (BatchNorm statistics)
    are updated by every replay, as in eager mode.  Gradients reach the module's parameters through ordinary AccumulateGrad."""

    def __init__(self, module, sample_args, warmup=3):
        if not all(isinstance(a, torch.Tensor) and a.is_cuda for a in sample_args):
            raise ValueError('sample_args must be CUDA tensors')
        from . import GRAPH_REPLAY_SAFE
        if (os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') != '0' or not GRAPH_REPLAY_SAFE) and os.environ.get('TAMTR_GRAPH_TIMING_ONLY') != '1':
            # (TAMTR_GRAPH_TIMING_ONLY=1: timing experiments with the runtime's packet capture on - gradients may be garbage)
            raise RuntimeError('GraphedPart needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP runtime starts (see the module docstring)')
        self.module = module
        self.names, self.params = zip(*[(n, p) for n, p in module.named_parameters()])
        alias = {n: torch.nn.Parameter(p.detach(), requires_grad=p.requires_grad) for n, p in zip(self.names, self.params)}  # same storage
        self.static_in = [a.detach().clone() for a in sample_args]
        self.stream = torch.cuda.Stream()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()

        def run():
            return torch.func.functional_call(module, alias, tuple(self.static_in))

        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):   # MIOpen / hipBLASLt solver selection, lazy workspaces, the leaves' accumulators: all before capture
                out = run()
                used = [(n, a) for n, a in alias.items() if a.requires_grad]
                g = torch.autograd.grad(out, [a for _, a in used], torch.ones_like(out), allow_unused=True)
                del out, g
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.fwd, pool=pool, stream=self.stream):
            self.static_out = run()
        self.static_gout = torch.zeros_like(self.static_out)
        leaves = [alias[n] for n in self.names if alias[n].requires_grad]
        with torch.cuda.graph(self.bwd, pool=pool, stream=self.stream):
            grads = torch.autograd.grad(self.static_out, leaves, self.static_gout, allow_unused=True)
        it = iter(grads)
        self.static_grads = [next(it) if alias[n].requires_grad else None for n in self.names]
        self.n_live = sum(g is not None for g in self.static_grads)
        self._probe = next(((p, g) for p, g in zip(self.params, self.static_grads) if g is not None), None)
        part = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *flat):
                for dst, src in zip(part.static_in, flat[:len(part.static_in)]):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                part.fwd.replay()
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

    def __call__(self, *args):
        if self._probe is not None and self._probe[0].grad is not None and self._probe[0].grad.data_ptr() == self._probe[1].data_ptr():
            # AccumulateGrad adopted the static gradient buffer as .grad last step; accumulating onto it would add the new
            # gradient to itself
            raise RuntimeError('GraphedPart: clear gradients with zero_grad(set_to_none=True) (or give .grad its own storage) between steps')
        return self._fn.apply(*args, *self.params)
