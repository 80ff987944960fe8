"""`Detect` - the yolo detection head API that north_star and SURVEY 8(b) list on the plugin surface
(ultralytics/nn/modules/head.py:22-82: `Detect(nc=80, ch=())`, `forward(x: list)`, `bias_init()`; DFL integral
nn/modules/block.py:17-36; anchor grid / ltrb decoding utils/tal.py:249-273).

TAMTR.yaml ends in ManbaWorldDecoder, not in Detect, so nothing here is on the benchmarked path: the class exists so that a
graph or checkpoint that names `Detect` builds and loads against this package - same constructor signature, same
state_dict keys (`cv2.{i}.{0,1}.conv/bn.*`, `cv2.{i}.2.{weight,bias}`, `cv3.*`, `dfl.conv.weight`), same outputs:
    train:  list of nl maps [B, 4*reg_max + nc, H_i, W_i]          (raw distribution logits | class logits)
    eval:   (y [B, 4 + nc, A], that list);  y = xywh in pixels | class probabilities, A = sum H_i W_i;  `export`: y alone
Branch convolutions are the trunk's Conv blocks (MIOpen + the BatchNorm/SiLU kernels of csrc/bn.hip in training).
"""
import math

import torch
import torch.nn as nn

from .backbone import Conv


class DFL(nn.Module):
    """Expectation of the per-side discrete distance distribution: softmax over `c1` bins dotted with 0..c1-1.  Kept as a frozen
    1x1 conv only for its state_dict key (`dfl.conv.weight`, requires_grad False, like the reference)."""

    def __init__(self, c1=16):
        super().__init__()
        self.c1 = c1
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        with torch.no_grad():
            self.conv.weight.copy_(torch.arange(c1, dtype=torch.float32).view(1, c1, 1, 1))

    def forward(self, x):
        """[B, 4*c1, A] -> [B, 4, A]."""
        b, _, a = x.shape
        p = x.view(b, 4, self.c1, a).softmax(2)
        return torch.einsum('bsca,c->bsa', p, self.conv.weight.view(self.c1).to(p.dtype))


def anchor_grid(maps, strides, offset=0.5):
    """Cell centres (in cells) and the stride of every anchor, level after level: ([A, 2] as (x, y), [A, 1])."""
    pts, st = [], []
    for f, s in zip(maps, strides):
        h, w = f.shape[-2:]
        ys = torch.arange(h, device=f.device, dtype=f.dtype) + offset
        xs = torch.arange(w, device=f.device, dtype=f.dtype) + offset
        pts.append(torch.stack([xs.repeat(h), ys.repeat_interleave(w)], 1))
        st.append(torch.full((h * w, 1), float(s), device=f.device, dtype=f.dtype))
    return torch.cat(pts), torch.cat(st)


class Detect(nn.Module):
    dynamic = False   # rebuild the anchor grid on every call
    export = False
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc, self.nl, self.reg_max = nc, len(ch), 16
        self.no = nc + 4 * self.reg_max
        self.stride = torch.zeros(self.nl)       # filled in by whoever builds the graph (tasks.py:283-290)
        c2, c3 = max(16, ch[0] // 4, 4 * self.reg_max), max(ch[0], min(nc, 100))

        def branch(c_in, c_mid, c_out):
            return nn.Sequential(Conv(c_in, c_mid, 3), Conv(c_mid, c_mid, 3), nn.Conv2d(c_mid, c_out, 1))
        self.cv2 = nn.ModuleList(branch(c, c2, 4 * self.reg_max) for c in ch)
        self.cv3 = nn.ModuleList(branch(c, c3, nc) for c in ch)
        self.dfl = DFL(self.reg_max)

    def forward(self, x):
        if len(x) != self.nl:
            raise ValueError(f'Detect was built for {self.nl} feature maps, got {len(x)}')
        first = x[0].shape
        for i in range(self.nl):                 # in place, like the reference: the caller's list ends up holding the raw maps
            x[i] = torch.cat([self.cv2[i](x[i]), self.cv3[i](x[i])], 1)
        if self.training:
            return x
        if self.dynamic or self.shape != first:
            pts, st = anchor_grid(x, self.stride)
            self.anchors, self.strides, self.shape = pts.t(), st.t(), first
        flat = torch.cat([m.flatten(2) for m in x], 2)                       # [B, no, A]
        ltrb = self.dfl(flat[:, :4 * self.reg_max])
        lo, hi = self.anchors[None] - ltrb[:, :2], self.anchors[None] + ltrb[:, 2:]
        y = torch.cat([(lo + hi) / 2 * self.strides, (hi - lo) * self.strides, flat[:, 4 * self.reg_max:].sigmoid()], 1)
        return y if self.export else (y, x)

    def bias_init(self):
        """Box-branch bias 1.0; class-branch bias = log-odds of ~5 objects per 640^2 image spread over nc classes at each stride."""
        for box, cls, s in zip(self.cv2, self.cv3, self.stride):
            box[-1].bias.data.fill_(1.0)
            cls[-1].bias.data[:self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)
