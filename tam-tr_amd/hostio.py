"""Sync-free host -> device staging for the small per-step metadata of the training objective (labels, denoising groups,
fixed match indices).

Why: `tensor.to('cuda')` from pageable memory ends in a stream synchronise (the reference does this several times per
step: nn/tasks.py:603-609, models/utils/ops.py:119,270-291), which stops the host thread from queueing step i+1 while the
GPU still runs the backward of step i.  On a step that issues ~5 800 kernels the host is the critical path for most of
the forward, so every such stall is wall time.  Here the bytes go through a small ring of pinned buffers and
`non_blocking=True` copies; a slot is reused only after the copy that last read it has completed (event per slot).
"""
import torch

_ALIGN = 64


class PinnedStager:
    def __init__(self, slots=4, nbytes=4 << 20):
        self.nbytes, self.nslots = nbytes, slots
        self.bufs, self.events = None, [None] * slots
        self.slot, self.cursor = 0, 0

    def next_step(self):
        """Call once at the top of a training step: rotate to the slot used `slots` steps ago."""
        self.slot = (self.slot + 1) % self.nslots
        self.cursor = 0
        ev = self.events[self.slot]
        if ev is not None:
            ev.synchronize()  # completed long ago unless the host is > slots-1 steps ahead

    def h2d(self, t, device, dtype=None):
        """CPU tensor -> device tensor without synchronising the stream (already-resident tensors pass through)."""
        device = torch.device(device)
        if dtype is not None:
            t = t.to(dtype)
        if t.device.type != 'cpu' or device.type == 'cpu':
            return t.to(device)
        n = t.numel() * t.element_size()
        if n == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=device)
        if self.bufs is None:
            self.bufs = [torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.nslots)]
        start = (self.cursor + _ALIGN - 1) // _ALIGN * _ALIGN
        if start + n > self.nbytes:  # oversized for the ring: ordinary (synchronising) copy, still correct
            return t.to(device)
        self.cursor = start + n
        view = self.bufs[self.slot][start:start + n].view(t.dtype).view(t.shape)
        view.copy_(t)
        out = view.to(device, non_blocking=True)
        if self.events[self.slot] is None:
            self.events[self.slot] = torch.cuda.Event()
        self.events[self.slot].record(torch.cuda.current_stream(device))
        return out


_STAGER = PinnedStager()


def stager():
    return _STAGER
