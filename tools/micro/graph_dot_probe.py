#!/usr/bin/env python3
"""What does hipGraphDebugDotPrint say about memset nodes?  (graphs.memset_nodes parses it.)"""
import os, re, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
x = torch.randn(8192, 512, device='cuda')
t = torch.ones(1 << 20, device='cuda')
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x.sum(0); t.zero_()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    y = x.sum(0)
    t.zero_()
    z = t + 1
path = tempfile.mktemp(suffix='.dot')
g.debug_dump(path)
txt = open(path, errors='replace').read() if os.path.exists(path) else None
print('dot file exists:', txt is not None, 'bytes:', 0 if txt is None else len(txt))
if txt:
    print(txt[:3000])
    print('...')
    print('labels:', sorted(set(re.findall(r'label="([^"\\]{0,40})', txt)))[:40])
from tamtr_amd.graphs import memset_nodes
g2 = torch.cuda.CUDAGraph(); g2.enable_debug_mode()
with torch.cuda.graph(g2, stream=s):
    y = x.sum(0); t.zero_(); z = t + 1
print('memset_nodes():', memset_nodes(g2))
g2.replay(); torch.cuda.synchronize(); print('replay after dump ok')
