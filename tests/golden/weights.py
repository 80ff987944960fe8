"""Deterministic, name-keyed weight filler shared by the golden-vector generator and the tests.

Golden fixtures do not store module weights (the full model is 42 M parameters).  Instead every
tensor of a ``state_dict`` is a pure function of (its key name, its shape, a base seed), so the
generator (which loads the values into the *reference* modules) and the tests (which load the same
values into the oracle / the HIP modules) agree without shipping the bytes.  ``checksum`` of the
filled dict is stored in each fixture to detect any drift of the torch CPU generator.

This file is test infrastructure: data plumbing only, no model arithmetic.
"""
import math
import zlib

import torch


def _gen(name, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) * 2654435761 + seed * 97 + 13) & 0x7FFFFFFF)
    return g


def tensor_for(name, shape, dtype=torch.float32, seed=0):
    """Value of state_dict entry `name` with `shape`."""
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    g = _gen(name, seed)
    if not dtype.is_floating_point:
        return torch.zeros(shape, dtype=dtype)  # num_batches_tracked
    rn = lambda: torch.randn(shape, generator=g, dtype=torch.float32)
    if leaf == 'running_var':
        t = 0.5 + torch.rand(shape, generator=g)
    elif leaf == 'running_mean':
        t = 0.1 * rn()
    elif leaf == 'logit_scale':
        t = math.log(1 / 0.07) + 0.05 * rn()
    elif leaf == 'A_logs':
        n = shape[-1]
        t = torch.log(torch.arange(1, n + 1, dtype=torch.float32)).expand(shape) + 0.05 * rn()
    elif leaf == 'Ds':
        t = 1.0 + 0.1 * rn()
    elif leaf == 'dt_projs_bias':
        t = -4.0 + 0.5 * rn()
    elif leaf == 'dt_projs_weight':
        t = rn() * shape[-1] ** -0.5
    elif leaf == 'x_proj_weight':
        t = rn() * shape[-1] ** -0.5
    elif leaf == 'bias' and shape == (1,) and 'score_head' in name:
        t = -10.0 + 0.1 * rn()  # ContrastiveHeadMLP.bias
    elif len(shape) <= 1 and leaf == 'weight':
        t = 1.0 + 0.1 * rn()  # BatchNorm / LayerNorm scale
    elif len(shape) <= 1:
        t = 0.1 * rn()  # every other bias-like vector
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = rn() / math.sqrt(fan_in)
    if 'bbox_head' in name and '.layers.2.' in name:
        # the reference zero-inits the last box-regression layer (head.py:1275-1281); keep it small so that boxes do
        # not saturate: 1 - sigmoid(z) loses all fp32 precision for z >~ 12 and inverse_sigmoid() then amplifies
        # rounding noise to percent level in the gradients (measured; see DESIGN.md "conditioning")
        t = t * 0.05
    return t.to(dtype).contiguous()


def fill_state(named_shapes, seed=0):
    """named_shapes: iterable of (name, shape, dtype) or a state_dict -> dict name -> tensor."""
    if isinstance(named_shapes, dict):
        named_shapes = [(k, v.shape, v.dtype) for k, v in named_shapes.items()]
    return {n: tensor_for(n, s, d, seed) for n, s, d in named_shapes}


def checksum(state):
    """Order-independent float64 checksum of a filled state dict."""
    tot = 0.0
    for k in sorted(state):
        v = state[k]
        if v.dtype.is_floating_point and v.numel():
            w = torch.arange(1, v.numel() + 1, dtype=torch.float64).remainder(7.0) + 1.0
            tot += float((v.double().flatten() * w).sum())
    return tot


def rnd(shape, seed, scale=1.0):
    """Seeded N(0, scale^2) input tensor (inputs too large to store are regenerated from their seed)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def urnd(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(shape, generator=g)


def summarize(t, full_max=4096, n_sample=1024):
    """Compact stand-in for a (possibly large) tensor: full copy if small, else strided sample + norms."""
    t = t.detach().to(torch.float32).contiguous().flatten()
    if t.numel() <= full_max:
        return {'full': t.numpy()}
    step = t.numel() // n_sample
    return {'sample': t[::step][:n_sample].clone().numpy(), 'step': step, 'l2': float(t.double().norm()),
            'sum': float(t.double().sum())}
