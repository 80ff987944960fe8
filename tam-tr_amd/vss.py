"""VMamba VSSBlock / SS2D for the MEH head with the external CUDA selective-scan extension REPLACED by our own gfx950
scan kernel (ops.selective_scan).  Reference: ultralytics/nn/extra_modules/VManba/vmamba.py:1169-1256 (VSSBlock),
:330-484,898-1038 (SS2D forward_type "v2"), csms6s.py:4-46 (CrossScan/CrossMerge), csms6s.py:252-270 (scan call).

state_dict keys are the reference's: norm, op.{x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, out_norm,
in_proj, conv2d, out_proj}, norm2, mlp.{fc1, fc2}.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import ops

_EINSUM_DT = os.environ.get('TAMTR_SS2D_EINSUM') == '1'


class TallLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose weight gradient on the GPU in bf16 is the split-K batched GEMM of
    ops.dw_splitk: the plain dW GEMM of a [B*H*W, C] activation runs on a handful of workgroups."""

    def forward(self, x):
        if x.is_cuda:
            if x.dtype == torch.bfloat16 or (torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16):
                x = x.to(torch.bfloat16)
            # (fp32 too: the Function's backward sums dW slices and the bias gradient with the ordered slab sum - autograd's own backward
            # of F.linear reduces with torch's multi-workgroup kernel, whose memset node does not replay under AQL packet capture)
            return ops.linear_splitk(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm semantics: keep w.p. 1-p, rescale by 1/(1-p)); identity in eval or p=0."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


def cross_scan(x):
    """[B,C,H,W] -> [B,4,C,HW]: row-major, column-major and both reversed (csms6s.py:4-14); autograd handles the merge."""
    a = x.flatten(2)
    b = x.transpose(2, 3).flatten(2)
    return torch.stack([a, b, a.flip(-1), b.flip(-1)], 1)


def cross_merge(ys, H, W):
    """[B,4,D,HW] -> [B,D,HW] (csms6s.py:26-34)."""
    B, K, D, L = ys.shape
    y = ys[:, 0:2] + ys[:, 2:4].flip(-1)
    return y[:, 0] + y[:, 1].view(B, D, W, H).transpose(2, 3).reshape(B, D, L)


class SS2D(nn.Module):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank='auto', d_conv=3, conv_bias=True, bias=False,
                 dt_min=0.001, dt_max=0.1, dt_scale=1.0, dt_init_floor=1e-4, **kwargs):
        super().__init__()
        d_inner = int(ssm_ratio * d_model)
        R = math.ceil(d_model / 16) if dt_rank == 'auto' else dt_rank
        K = 4
        self.d_inner, self.dt_rank, self.d_state = d_inner, R, d_state
        self.out_norm = nn.LayerNorm(d_inner)
        self.in_proj = TallLinear(d_model, 2 * d_inner, bias=bias)
        self.conv2d = nn.Conv2d(d_inner, d_inner, d_conv, padding=(d_conv - 1) // 2, groups=d_inner, bias=conv_bias)
        self.x_proj_weight = nn.Parameter(torch.stack([nn.Linear(d_inner, R + 2 * d_state, bias=False).weight.detach()
                                                       for _ in range(K)], 0))
        self.out_proj = TallLinear(d_inner, d_model, bias=bias)
        # dt projection init (vmamba.py:152-176): weight U(-R^-0.5, R^-0.5), bias = softplus^-1(dt), dt log-uniform
        std = R ** -0.5 * dt_scale
        self.dt_projs_weight = nn.Parameter(torch.empty(K, d_inner, R).uniform_(-std, std))
        dt = torch.exp(torch.rand(K, d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        self.dt_projs_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.A_logs = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(K * d_inner, 1))
        self.Ds = nn.Parameter(torch.ones(K * d_inner))
        self.A_logs._no_weight_decay = True
        self.Ds._no_weight_decay = True

    def forward(self, x):  # x: [B,H,W,C]
        B, H, W, _ = x.shape
        xz = self.in_proj(x)
        xi, z = xz.chunk(2, -1)
        K, R, N, L, D = 4, self.dt_rank, self.d_state, H * W, self.d_inner
        fused_front = not _EINSUM_DT and D % 32 == 0 and self.conv2d.kernel_size == (3, 3)
        xc = None if fused_front else self.conv2d(xi.permute(0, 3, 1, 2).contiguous())  # [B,D,H,W]
        # Cross-scan WITHOUT materialising the four sequences (csms6s.py:4-14): directions 0/2 walk the row-major flattening
        # forwards/backwards, 1/3 the column-major one; the kernel reads the two stored copies and reverses on the fly, and all
        # per-direction operands are kept in the un-reversed order of their base copy.  The scan runs in fp32 (vmamba.py:980).
        if fused_front and x.is_cuda and D in (64, 128, 256, 512, 1024) and os.environ.get('TAMTR_SS2D_SPLIT') != '1':
            # the whole core as one autograd node (ops._SS2DCore): same kernels, planned backward buffers
            As = -torch.exp(self.A_logs.float())
            g = ops.ss2d_core(xz, self.conv2d.weight, self.conv2d.bias, self.x_proj_weight, self.dt_projs_weight.float().reshape(K * D, R), As,
                              self.Ds.float(), self.dt_projs_bias.float().reshape(-1), self.out_norm.weight, self.out_norm.bias,
                              self.out_norm.eps, R, N)
            return self.out_proj(g.view(B, H, W, D))
        if _EINSUM_DT:
            xi = F.silu(xc).float()
            u2 = torch.stack([xi.flatten(2), xi.transpose(2, 3).flatten(2)], 1)  # [B,2,D,L]
            wx = self.x_proj_weight.float()  # [4, R+2N, D]
            xd_a = torch.matmul(torch.cat([wx[0], wx[2]], 0), u2[:, 0])  # [B, 2C, L]: directions 0 and 2 (same base order)
            xd_b = torch.matmul(torch.cat([wx[1], wx[3]], 0), u2[:, 1])  # directions 1 and 3
            C = R + 2 * N
            x_dbl = torch.stack([xd_a[:, :C], xd_b[:, :C], xd_a[:, C:], xd_b[:, C:]], 1)  # [B,4,C,L]
            dtr, Bs, Cs = (t.contiguous() for t in torch.split(x_dbl, [R, N, N], 2))
        else:
            if fused_front:  # depthwise conv + SiLU + both flattenings in one kernel, read from the channels-last in_proj output
                u2 = ops.dwconv_silu_cross(xz, self.conv2d.weight, self.conv2d.bias, D)
            else:
                u2 = ops.cross_scan_input(xc)                               # SiLU + both flattenings, [B,2,D,L] fp32
            dtr, Bs, Cs = ops.x_proj_cross(self.x_proj_weight, u2, R, N)    # [B,4,R|N|N,L]
        # the dt projection (einsum "bkrl,kdr->bkdl", vmamba.py:972) happens INSIDE the scan kernels: the [B, 4*d_inner, L]
        # delta tensor is never written, and its skinny K = R <= 32 GEMMs (forward + two backward: ~38 ms per step through
        # rocBLAS at these shapes) disappear
        As = -torch.exp(self.A_logs.float())
        if _EINSUM_DT:  # A/B switch (env TAMTR_SS2D_EINSUM=1): reference-shaped einsum + materialised delta
            dts = torch.einsum('bkrl,kdr->bkdl', dtr, self.dt_projs_weight.float())
            ys = ops.selective_scan_cross_delta(u2, dts.reshape(B, -1, L), As, Bs, Cs, self.Ds.float(),
                                                self.dt_projs_bias.float().reshape(-1)).view(B, K, D, L)
            # cross-merge (csms6s.py:26-34) on un-reversed outputs: no flips left
            y = ys[:, 0] + ys[:, 2] + (ys[:, 1] + ys[:, 3]).view(B, D, W, H).transpose(2, 3).reshape(B, D, L)
        else:  # scan + cross-merge as one autograd node (the merged gradient feeds the scan backward directly)
            fused_back = D in (64, 128, 256, 512, 1024)
            y = ops.selective_scan_cross_merged(u2, dtr, self.dt_projs_weight.float().reshape(K * D, R), As, Bs, Cs, self.Ds.float(),
                                                self.dt_projs_bias.float().reshape(-1), H, W, token_major=fused_back)
            if fused_back:  # y is [B, L, D]: out_norm + SiLU(z) gate in one kernel, z read from xz where it lies
                g = ops.ln_gate(y, xz, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps)
                return self.out_proj(g.view(B, H, W, D))
        y = self.out_norm(y.transpose(1, 2)).view(B, H, W, -1)
        return self.out_proj((y * F.silu(z)).to(x.dtype))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = TallLinear(in_features, hidden_features or in_features)
        self.fc2 = TallLinear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class VSSBlock(nn.Module):
    """x + DropPath(SS2D(LN(x))); x + DropPath(Mlp(LN(x))), channels-last."""

    def __init__(self, hidden_dim=0, drop_path=0.0, ssm_d_state=16, ssm_ratio=2.0, ssm_dt_rank='auto', ssm_conv=3,
                 ssm_conv_bias=True, mlp_ratio=4.0, **kwargs):
        super().__init__()
        self.norm = nn.LayerNorm(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank, d_conv=ssm_conv,
                       conv_bias=ssm_conv_bias)
        self.drop_path = DropPath(drop_path)
        self.norm2 = nn.LayerNorm(hidden_dim)
        self.mlp = Mlp(hidden_dim, int(hidden_dim * mlp_ratio))

    @staticmethod
    def _ln(norm, x):
        if x.is_cuda and x.shape[-1] in (32, 64, 128, 256, 512, 1024) and x.dtype in (torch.float32, torch.bfloat16):
            return ops.layer_norm(x, norm.weight, norm.bias, norm.eps)  # activation dtype in and out (no autocast casts around it)
        return norm(x)

    def _residual(self, x, branch, scale=None):
        """x + DropPath(branch) in one kernel: the per-sample keep mask / (1 - p) is a [B,1,1,1] factor of an addcmul (timm's
        DropPath as separate ops is a mul, a div and an add over the whole map).  `scale` [B]: the factor drawn by the caller
        (draw_drop_scales) - a recorded HIP graph takes it as an INPUT, so a replay and an eager pass given the same draw are the same
        function (the draw itself stays outside the graph, on the ordinary generator)."""
        p = self.drop_path.drop_prob
        if scale is not None:
            return torch.addcmul(x, branch, scale.to(x.dtype).view((x.shape[0],) + (1,) * (x.dim() - 1)))
        if p == 0.0 or not self.training:
            return x + branch
        keep = 1.0 - p
        scale = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep) / keep
        return torch.addcmul(x, branch, scale)

    def draw_drop_scales(self, n, device):
        """[2, n] fp32: keep mask / (1 - p) of the block's two residual branches (ones when DropPath is inactive)."""
        p = self.drop_path.drop_prob
        if p == 0.0 or not self.training:
            return torch.ones(2, n, device=device)
        keep = 1.0 - p
        return torch.empty(2, n, device=device).bernoulli_(keep) / keep

    def forward(self, x, drop_scales=None):
        x = x.contiguous()  # the head hands in a permuted NCHW view: one copy here keeps both residual adds contiguous
        s1, s2 = (None, None) if drop_scales is None else (drop_scales[0], drop_scales[1])
        x = self._residual(x, self.op(self._ln(self.norm, x)), s1)
        return self._residual(x, self.mlp(self._ln(self.norm2, x)), s2)
