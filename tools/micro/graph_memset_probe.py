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

# 4. memset nodes on REUSED pool memory (what a recorded training step is made of): is a memset node ordered against the kernels
#    before and after it?  Chain per link: kernel writes ones into a block -> block freed -> same block re-allocated -> memset to zero
#    -> kernel adds 1 in place -> result copied out.  Right answer: every element 1.  A memset executed before the ones-kernel of its
#    own link (or not at all) leaves 2; one executed after the add leaves 0.
N, LINKS = 1 << 18, 40
outs = torch.empty(LINKS, N, device=dev)


def chain():
    ptrs = set()
    for i in range(LINKS):
        a = torch.empty(N, device=dev)
        a.fill_(1.0)                 # kernel
        ptrs.add(a.data_ptr())
        del a
        b = torch.empty(N, device=dev)
        ptrs.add(b.data_ptr())
        b.zero_()                    # memset node
        b += 1.0                     # kernel
        outs[i].copy_(b)             # kernel
        del b
    return len(ptrs)
g, nptr = capture(chain)
hist = []
for rep in range(4):
    outs.fill_(-7.0)
    g.replay()
    torch.cuda.synchronize()
    vals, counts = torch.unique(outs, return_counts=True)
    hist.append({str(float(v)): int(c) for v, c in zip(vals, counts)})
res['reused_block_chain'] = {'distinct_blocks': nptr, 'links': LINKS, 'elements_per_link': N, 'value_histogram_per_replay': hist}

# 5. the same with a torch global reduction in the chain (its semaphore memset lands on reused memory)
x = torch.randn(8192, 512, device=dev)
sums = torch.empty(LINKS, 512, device=dev)


def chain2():
    for i in range(LINKS):
        a = torch.empty(1 << 16, device=dev)
        a.fill_(3.0)                 # dirties the block the reduction's semaphores / staging buffer will get
        del a
        sums[i].copy_(x.sum(0))
g, _ = capture(chain2)
errs = []
for rep in range(4):
    x.copy_(torch.randn(8192, 512, device=dev))
    sums.fill_(float('nan'))
    g.replay()
    torch.cuda.synchronize()
    ref = x.double().sum(0)
    e = (sums.double() - ref).norm(dim=1) / ref.norm()
    errs.append({'links_off': int((~(e < 1e-5)).sum()), 'worst_rel': float(torch.nan_to_num(e, nan=1e30).max())})
res['reduction_on_reused_blocks'] = errs
print(json.dumps({k: res[k] for k in ('packet_capture', 'reused_block_chain', 'reduction_on_reused_blocks')}), flush=True)
