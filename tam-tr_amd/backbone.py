"""GELAN backbone / neck blocks that surround the hot path ("rest PyTorch-ROCm", BASELINE configs[1]).

These are stock PyTorch modules (MIOpen convolutions) - they are NOT part of the hand-written HIP path (except CPAM, whose
gates run in the fused kernels of csrc/cpam.hip); they exist so that the TAMTR graph can be assembled and its checkpoints (state_dict keys) stay interchangeable with the reference:
  Conv            ultralytics/nn/modules/conv.py:23-40
  RepConvN ... RepNCSPELAN4, SPPELAN, CPAM   ultralytics/nn/extra_modules/block.py:26-163,255-308
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

def _conv(conv, x):
    """conv(x); on the GPU through ops.conv2d_module (1x1 convolutions: weight gradient off MIOpen's memset + atomic solvers)."""
    if x.is_cuda:
        from . import ops
        return ops.conv2d_module(conv, x)
    return conv(x)


BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # the reference rewrites every BatchNorm2d after construction (torch_utils.py:303-313)


def batchnorm(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOMENTUM)


def fold_bn(conv, bn):
    """conv followed by BatchNorm with frozen statistics == one biased conv (ultralytics/utils/torch_utils.py:159-180).
    Returned frozen (requires_grad False), as the reference's fused evaluation graph holds it."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups,
                      bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
        fused.weight.copy_((scale[:, None] * conv.weight.reshape(conv.out_channels, -1)).view(fused.weight.shape))
        b = torch.zeros(conv.out_channels, device=conv.weight.device) if conv.bias is None else conv.bias
        fused.bias.copy_(scale * b + (bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))))
    return fused


class Conv(nn.Module):
    """conv (no bias, 'same' padding) -> BatchNorm -> SiLU | identity.  Args as the reference: (c1, c2, k, s, p, g, d, act).
    After fuse() the BatchNorm is folded into the conv (evaluation graph, nn/tasks.py:121-152)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if p is None:
            p = (d * (k - 1) + 1) // 2 if d > 1 else k // 2
        self.conv = nn.Conv2d(c1, c2, k, s, p, dilation=d, groups=g, bias=False)
        self.bn = batchnorm(c2)
        self.act = nn.SiLU() if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    def fuse(self):
        if 'bn' in self._modules:
            self.conv = fold_bn(self.conv, self.bn)
            del self.bn

    def fusable(self, y):
        """The BatchNorm (+ SiLU) of this block can run on csrc/bn.hip for the convolution output y."""
        return ('bn' in self._modules and y.is_cuda and self.bn.training and self.bn.affine and type(self.act) in (nn.SiLU, nn.Identity)
                and y.dtype in (torch.bfloat16, torch.float32))

    def forward(self, x, residual=None):
        """residual: added to the block's output (the shortcut of a bottleneck); on the channels-last kernel path it joins inside the
        BatchNorm apply pass."""
        return self.post(_conv(self.conv, x), residual)

    def post(self, y, residual=None):
        """Everything after the convolution: BatchNorm, activation, optional shortcut."""
        if 'bn' not in self._modules:
            y = self.act(y)
            return y if residual is None else y + residual
        if self.fusable(y):
            from . import ops
            if ops.is_cl(y) and ops.bn_cl_ok(y.shape[1], y.dtype):
                B, C, H, W = y.shape  # channels-last map: the [B*H*W, C] kernels, result stays channels-last
                res = None
                if residual is not None and ops.is_cl(residual) and residual.shape == y.shape:
                    res, residual = residual.permute(0, 2, 3, 1).reshape(B * H * W, C), None
                o = ops.bn_act(y.permute(0, 2, 3, 1).reshape(B * H * W, C), self.bn, isinstance(self.act, nn.SiLU), res)
                o = o.view(B, H, W, C).permute(0, 3, 1, 2)
                return o if residual is None else o + residual
            y = ops.bn_act(y, self.bn, isinstance(self.act, nn.SiLU))  # BatchNorm (batch stats) + SiLU: csrc/bn.hip
        else:
            y = self.act(self.bn(y))
        return y if residual is None else y + residual


class RepConvN(nn.Module):
    """Training-time RepVGG pair: SiLU(3x3 conv+BN  +  1x1 conv+BN); switch_to_deploy() merges the pair into one biased 3x3
    conv for evaluation (ultralytics/nn/extra_modules/block.py:53-124)."""

    def __init__(self, c1, c2, k=3, s=1, p=1, g=1, d=1, act=True, bn=False, deploy=False):
        super().__init__()
        assert k == 3 and p == 1
        self.conv1 = Conv(c1, c2, 3, s, p=1, g=g, act=False)
        self.conv2 = Conv(c1, c2, 1, s, p=0, g=g, act=False)
        self.act = nn.SiLU() if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    def forward(self, x):
        if 'conv' in self._modules:
            return self.act(self.conv(x))
        c1, c2 = self.conv1, self.conv2
        if x.is_cuda and type(self.act) in (nn.SiLU, nn.Identity) and 'bn' in c1._modules and 'bn' in c2._modules:
            from . import ops
            y1, y2 = c1.conv(x), _conv(c2.conv, x)
            if (c1.fusable(y1) and c2.fusable(y2) and ops.is_cl(y1) and ops.is_cl(y2) and ops.bn_cl_ok(y1.shape[1], y1.dtype)
                    and c1.bn.track_running_stats and c2.bn.track_running_stats and c1.bn.eps == c2.bn.eps and c1.bn.momentum == c2.bn.momentum):
                B, C, H, W = y1.shape   # both BatchNorms, the sum and the activation in one pass each way (csrc/bn.hip bncl2_*)
                o = ops.bn2_act(y1.permute(0, 2, 3, 1).reshape(B * H * W, C), c1.bn, y2.permute(0, 2, 3, 1).reshape(B * H * W, C), c2.bn,
                                isinstance(self.act, nn.SiLU))
                return o.view(B, H, W, C).permute(0, 3, 1, 2)
            return self.act(c1.post(y1) + c2.post(y2))
        return self.act(self.conv1(x) + self.conv2(x))

    @staticmethod
    def _branch(branch):
        bn = branch.bn
        std = (bn.running_var + bn.eps).sqrt()
        return branch.conv.weight * (bn.weight / std).reshape(-1, 1, 1, 1), bn.bias - bn.running_mean * bn.weight / std

    def switch_to_deploy(self):
        if 'conv' in self._modules:
            return
        with torch.no_grad():
            (k3, b3), (k1, b1) = self._branch(self.conv1), self._branch(self.conv2)
            c = self.conv1.conv
            self.conv = nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.groups,
                                  bias=True).requires_grad_(False).to(c.weight.device)
            self.conv.weight.copy_(k3 + F.pad(k1, [1, 1, 1, 1]))
            self.conv.bias.copy_(b3 + b1)
        del self.conv1, self.conv2


class RepNBottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = RepConvN(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        if self.add:
            return self.cv2(self.cv1(x), residual=x)   # x + cv2(cv1(x)), the shortcut folded into cv2's BatchNorm pass
        return self.cv2(self.cv1(x))


class RepNCSP(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, c_, 1, 1), Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(RepNBottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)))

    def forward(self, x):
        from . import ops
        return self.cv3(ops.cat_channels((self.m(self.cv1(x)), self.cv2(x))))


class RepNCSPELAN4(nn.Module):
    """GELAN block: 1x1 split, two (RepNCSP -> 3x3) stages, concat of all four branches, 1x1 fuse."""

    def __init__(self, c1, c2, c3, c4, c5=1):
        super().__init__()
        self.c = c3 // 2
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv2 = nn.Sequential(RepNCSP(c3 // 2, c4, c5), Conv(c4, c4, 3, 1))
        self.cv3 = nn.Sequential(RepNCSP(c4, c4, c5), Conv(c4, c4, 3, 1))
        self.cv4 = Conv(c3 + 2 * c4, c2, 1, 1)

    def branches(self, x):
        from . import ops
        y = list(ops.chunk2_channels(self.cv1(x)))
        y.append(self.cv2(ops.pack_channels(y[-1])))   # one packed copy of the half for the two convolutions that read it
        y.append(self.cv3(y[-1]))
        return y

    def forward(self, x):
        from . import ops
        return self.cv4(ops.cat_channels(self.branches(x)))


class SPPELAN(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.c = c3
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv5 = Conv(4 * c3, c2, 1, 1)

    def forward(self, x):
        from . import ops
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(ops.max_pool2d(y[-1], 5, 1, 2) if y[-1].is_cuda else F.max_pool2d(y[-1], 5, 1, 2))
        return self.cv5(ops.cat_channels(y))


class CPAM(nn.Module):
    """Parameter-free channel + spatial max gates (extra_modules/block.py:271-308)."""

    def __init__(self, c1, c2=None):
        super().__init__()

    def forward(self, x):
        from . import ops
        return ops.cpam(x)  # fused HIP kernels; no torch fallback


class Upsample(nn.Upsample):
    """nn.Upsample (TAMTR.yaml uses nearest with scale 2.0 and 0.5).  For exactly those two factors nearest resampling is an index
    pattern: x[..., ::2, ::2] and a 2x2 repeat.  Under autocast F.interpolate is an fp32 op: the [16,128,320,320] map of layer 29
    was cast to fp32 (840 MB) just to be subsampled."""

    def forward(self, x):
        if self.mode == 'nearest' and x.dim() == 4 and self.size is None and isinstance(self.scale_factor, (int, float)):
            if x.is_cuda and self.scale_factor in (0.5, 2.0):   # channels-last map: one kernel each way (csrc/layout.hip)
                from . import ops
                y = ops.resample2(x, self.scale_factor == 2.0)
                if y is not None:
                    return y
            if self.scale_factor == 0.5 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                return x[..., ::2, ::2]
            if self.scale_factor == 2.0:
                B, C, H, W = x.shape
                if not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):  # stays channels-last
                    y = x.permute(0, 2, 3, 1)[:, :, None, :, None, :].expand(B, H, 2, W, 2, C).reshape(B, 2 * H, 2 * W, C)
                    return y.permute(0, 3, 1, 2)
                return x[:, :, :, None, :, None].expand(B, C, H, 2, W, 2).reshape(B, C, 2 * H, 2 * W)
        return super().forward(x)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        if self.d == 1:
            from . import ops
            return ops.cat_channels(x)
        return torch.cat(x, self.d)
