"""state_dict layout (key names, shapes, dtypes) of every module on the hot path - TEST INFRASTRUCTURE ONLY.

The key names are the reference's checkpoint surface (SURVEY.md 8b "state_dict key names").  Each fixture stores a
checksum of the name-keyed weights that were loaded into the *reference* module; tests rebuild the same dict from
these specs, so a wrong key name or shape anywhere here (or in the product modules, which are checked against
these specs too) changes the checksum and fails the test.
"""
import math

import torch

F32, I64 = torch.float32, torch.int64


def _pre(prefix, items):
    return [(prefix + n, s, d) for n, s, d in items]


def bn(c):
    return [('weight', (c,), F32), ('bias', (c,), F32), ('running_mean', (c,), F32), ('running_var', (c,), F32),
            ('num_batches_tracked', (), I64)]


def conv(c1, c2, k=1, g=1):
    return [('conv.weight', (c2, c1 // g, k, k), F32)] + _pre('bn.', bn(c2))


def linear(i, o, bias=True):
    return [('weight', (o, i), F32)] + ([('bias', (o,), F32)] if bias else [])


def layernorm(c):
    return [('weight', (c,), F32), ('bias', (c,), F32)]


def repncsp(c1, c2):
    c_ = c2 // 2
    return (_pre('cv1.', conv(c1, c_)) + _pre('cv2.', conv(c1, c_)) + _pre('cv3.', conv(2 * c_, c2)) +
            _pre('m.0.cv1.conv1.', conv(c_, c_, 3)) + _pre('m.0.cv1.conv2.', conv(c_, c_, 1)) +
            _pre('m.0.cv2.', conv(c_, c_, 3)))


def elan(c1, c2, c3, c4):
    return (_pre('cv1.', conv(c1, c3)) + _pre('cv2.0.', repncsp(c3 // 2, c4)) + _pre('cv2.1.', conv(c4, c4, 3)) +
            _pre('cv3.0.', repncsp(c4, c4)) + _pre('cv3.1.', conv(c4, c4, 3)) + _pre('cv4.', conv(c3 + 2 * c4, c2)))


def gate(c, nh, gc=512):
    return [('bias', (nh,), F32)] + _pre('gl.', linear(gc, c)) + _pre('proj_conv.', conv(c, c, 3))


def tiagelan(c1, c2, c3, c4, nh):
    return elan(c1, c2, c3, c4) + _pre('attn.', gate(c4, nh))


def sppelan(c1, c2, c3):
    return _pre('cv1.', conv(c1, c3)) + _pre('cv5.', conv(4 * c3, c2))


def msdeform(d, nl, nh, npts):
    return (_pre('sampling_offsets.', linear(d, nh * nl * npts * 2)) + _pre('attention_weights.', linear(d, nh * nl * npts)) +
            _pre('value_proj.', linear(d, d)) + _pre('output_proj.', linear(d, d)))


def decoder_layer(d, nh, ffn, nl, npts=4):
    return ([('self_attn.in_proj_weight', (3 * d, d), F32), ('self_attn.in_proj_bias', (3 * d,), F32)] +
            _pre('self_attn.out_proj.', linear(d, d)) + _pre('norm1.', layernorm(d)) +
            _pre('cross_attn.', msdeform(d, nl, nh, npts)) + _pre('norm2.', layernorm(d)) +
            _pre('linear1.', linear(d, ffn)) + _pre('linear2.', linear(ffn, d)) + _pre('norm3.', layernorm(d)))


def mlp(i, h, o, n):
    dims = [i] + [h] * (n - 1) + [o]
    out = []
    for k in range(n):
        out += _pre(f'layers.{k}.', linear(dims[k], dims[k + 1]))
    return out


def contrastive():
    return [('bias', (1,), F32), ('logit_scale', (), F32)]


def vss_block(d, d_state=16, ssm_ratio=2.0, mlp_ratio=4.0):
    di, R = int(ssm_ratio * d), math.ceil(d / 16)
    op = ([('x_proj_weight', (4, R + 2 * d_state, di), F32), ('dt_projs_weight', (4, di, R), F32),
           ('dt_projs_bias', (4, di), F32), ('A_logs', (4 * di, d_state), F32), ('Ds', (4 * di,), F32)] +
          _pre('out_norm.', layernorm(di)) + _pre('in_proj.', linear(d, 2 * di, False)) +
          [('conv2d.weight', (di, 1, 3, 3), F32), ('conv2d.bias', (di,), F32)] + _pre('out_proj.', linear(di, d, False)))
    h = int(d * mlp_ratio)
    return (_pre('norm.', layernorm(d)) + _pre('op.', op) + _pre('norm2.', layernorm(d)) +
            _pre('mlp.fc1.', linear(d, h)) + _pre('mlp.fc2.', linear(h, d)))


def text_decoder_heads(hd, nh, ffn, ndl, nl=3):
    out = []
    for i in range(ndl):
        out += _pre(f'decoder.layers.{i}.', decoder_layer(hd, nh, ffn, nl))
    for i in range(ndl):
        out += _pre(f'dec_bbox_head.{i}.', mlp(hd, hd, 4, 3))
    for i in range(ndl):
        out += _pre(f'dec_score_head.{i}.', contrastive())
    return out + _pre('query_pos_head.', mlp(4, 2 * hd, hd, 2))


def meh_head(nc, ch, hd, nh, ndl, ffn, vss=True, dims=None):
    out = []
    for i, c in enumerate(ch):
        out += [(f'input_proj.{i}.0.weight', (hd, c, 1, 1), F32)] + _pre(f'input_proj.{i}.1.', bn(hd))
    if vss:
        for i, c in enumerate(dims or ch):
            out += _pre(f'VSSBlocks.{i}.', vss_block(c))
    out += text_decoder_heads(hd, nh, ffn, ndl, len(ch))
    out += [('denoising_class_embed.weight', (nc + 1, hd), F32)]
    out += _pre('enc_output.0.', linear(hd, hd)) + _pre('enc_output.1.', layernorm(hd))
    out += _pre('enc_score_head.', linear(hd, nc)) + _pre('enc_bbox_head.', mlp(hd, hd, 4, 3))
    return out


def tamtr_model(nc=10, vss=True):
    """Full graph of cfg/models/TAMTR/TAMTR.yaml -> 'model.{i}.' keys."""
    L = {0: conv(3, 64, 3), 1: conv(64, 128, 3), 2: elan(128, 256, 128, 64), 3: conv(256, 256, 3),
         4: elan(256, 512, 256, 128), 5: conv(512, 512, 3), 6: elan(512, 512, 512, 256), 7: conv(512, 512, 3),
         8: elan(512, 512, 512, 256), 9: sppelan(512, 512, 256), 10: conv(512, 512), 12: conv(512, 512),
         13: conv(512, 512), 16: tiagelan(1536, 512, 512, 256, 8), 18: conv(512, 256), 20: conv(512, 256),
         21: conv(256, 256), 24: tiagelan(768, 256, 256, 128, 4), 26: conv(256, 128), 28: conv(256, 128),
         29: conv(64, 128), 32: tiagelan(384, 128, 128, 64, 2), 34: conv(128, 128, 3),
         36: tiagelan(384, 256, 256, 128, 4), 38: conv(256, 256, 3), 40: tiagelan(768, 512, 512, 256, 8),
         41: meh_head(nc, [128, 256, 512], 512, 8, 3, 1024, vss)}
    out = []
    for i in sorted(L):
        out += _pre(f'model.{i}.', L[i])
    return out
