#!/usr/bin/env python3
"""Do hipMemsetAsync nodes and torch's multi-block reductions survive a HIP-graph replay under AQL packet capture of graph nodes?

    PACKET_CAPTURE=0|1 python3 tools/micro/graph_memset_probe.py

torch's global reduction (Reduce.cuh: several workgroups per output, the last one to arrive finishes) counts arrivals in a semaphore buffer
that it zeroes with cudaMemsetAsync before EVERY launch and never resets itself.  If a replay drops or mis-replays that tiny memset node,
the first replay still works (fresh zeros) and every later one leaves its output unwritten."""
import json, os
os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = os.environ.get('PACKET_CAPTURE', '0')
import torch

dev = 'cuda'
res = {'packet_capture': os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'], 'hip': torch.version.hip}
s = torch.cuda.Stream()


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = fn()
    return g, out


# 1. memset nodes of several sizes (zero_() of a contiguous tensor is cudaMemsetAsync)
memset = {}
for nbytes in (4, 64, 256, 1024, 4096, 1 << 16, 1 << 20, 1 << 26):
    t = torch.ones(nbytes, dtype=torch.uint8, device=dev)
    g, _ = capture(lambda: t.zero_())
    bad = []
    for rep in range(4):
        t.fill_(1 + rep)
        g.replay()
        torch.cuda.synchronize()
        bad.append(int((t != 0).sum()))
    memset[nbytes] = bad
res['memset_nonzero_bytes_after_replay'] = memset

# 2. a multi-block torch reduction (column sums of a tall matrix: the shape of the kernels' partial-sum buffers)
red = {}
for shape in ((512, 2, 256), (4096, 512), (64, 16, 2560), (100000,)):
    x = torch.randn(*shape, device=dev)
    g, y = capture(lambda: x.sum(0))
    errs = []
    for rep in range(4):
        x.copy_(torch.randn(*shape, device=dev))
        g.replay()
        torch.cuda.synchronize()
        ref = x.double().sum(0)
        errs.append(float((y.double() - ref).norm() / ref.norm().clamp_min(1e-30)))
    red[str(shape)] = errs
res['sum0_rel_err_per_replay'] = red

# 3. torch.zeros inside the capture, then an accumulation into it
a = torch.randn(1 << 20, device=dev)


def zacc():
    z = torch.zeros_like(a)
    z += a
    return z
g, z = capture(zacc)
errs = []
for rep in range(4):
    a.copy_(torch.randn(1 << 20, device=dev))
    g.replay()
    torch.cuda.synchronize()
    errs.append(float((z - a).abs().max()))
res['zeros_then_add_max_err'] = errs
print(json.dumps(res), flush=True)
