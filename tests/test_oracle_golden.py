"""CPU: pin the oracle (oracle/tamtr_oracle.py) against the reference-generated golden vectors (tests/golden/*.npz).

Each test rebuilds the name-keyed weights from oracle/specs.py, checks their checksum against the one recorded when
the same weights were loaded into the reference module (=> key names and shapes are the reference's), then compares
outputs and gradients.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T, assert_close, check_param_grads, check_summary
from oracle import specs
from oracle import tamtr_oracle as O
from weights import checksum, fill_state, rnd, urnd

RT, AT = 2e-4, 2e-5


def make_state(spec, seed, wsum=None, grad=True):
    st = fill_state(spec, seed)
    if wsum is not None:
        assert abs(checksum(st) - float(wsum)) <= 1e-6 * max(1.0, abs(float(wsum))), 'state_dict layout differs from reference'
    if grad:
        for k, v in st.items():
            if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_()
    return st


def grads(st):
    return {k: v.grad for k, v in st.items() if v.requires_grad}


@pytest.mark.parametrize('tag', ['A', 'B', 'C'])
def test_gate(golden, tag):
    fx = golden('gate')
    B, c, nh, H, W, Tn, train = [int(v) for v in fx[f'{tag}.cfg']]
    st = make_state(specs.gate(c, nh), 11, fx[f'{tag}.wsum'])
    x, g = T(fx[f'{tag}.x']).requires_grad_(), T(fx[f'{tag}.guide']).requires_grad_()
    out = O.maxsigmoid_attn_block(x, g, O.View(st), nh, bool(train))
    assert_close(out, fx[f'{tag}.out'], RT, AT, 'gate out')
    if train:
        (out * T(fx[f'{tag}.cot'])).sum().backward()
        check_summary(fx, f'{tag}.gin.x', x.grad, RT, AT)
        check_summary(fx, f'{tag}.gin.guide', g.grad, RT, AT)
        check_param_grads(fx, f'{tag}.', grads(st), RT, AT)
        assert_close(st['proj_conv.bn.running_mean'], fx[f'{tag}.bn_mean'], RT, AT, 'bn mean')
        assert_close(st['proj_conv.bn.running_var'], fx[f'{tag}.bn_var'], RT, AT, 'bn var')


def test_tiagelan(golden):
    fx = golden('tiagelan')
    st = make_state(specs.tiagelan(96, 64, 128, 64, 2), 12, fx['wsum'])
    x, g = T(fx['x']).requires_grad_(), T(fx['guide']).requires_grad_()
    out = O.tiagelan(x, g, O.View(st), 2, True)
    assert_close(out, fx['out'], RT, AT, 'tiagelan out')
    (out * T(fx['cot'])).sum().backward()
    check_summary(fx, 'gin.x', x.grad, RT, AT)
    assert g.grad is None  # the gate result is discarded, no gradient reaches the text
    gr = grads(st)
    check_param_grads(fx, '', gr, RT, AT)
    assert all(gr[k] is None for k in gr if k.startswith('attn.'))
    assert_close(st['attn.proj_conv.bn.running_mean'], fx['attn_bn_mean'], RT, AT, 'attn bn mean (side effect kept)')
    assert_close(st['attn.proj_conv.bn.running_var'], fx['attn_bn_var'], RT, AT)
    with torch.no_grad():
        assert_close(O.tiagelan(x, g, O.View(st), 2, False), fx['out_eval'], RT, AT, 'eval')


@pytest.mark.parametrize('tag', ['A', 'B', 'C'])
def test_msdeform_core(golden, tag):
    fx = golden('msdeform_core')
    v, loc, aw = (T(fx[f'{tag}.{k}']).requires_grad_() for k in ('value', 'loc', 'aw'))
    out = O.ms_deform_attn_core(v, fx[f'{tag}.shapes'].tolist(), loc, aw)
    assert_close(out, fx[f'{tag}.out'], RT, AT, 'core out')
    (out * T(fx[f'{tag}.cot'])).sum().backward()
    assert_close(v.grad, fx[f'{tag}.g_value'], RT, AT, 'g_value')
    assert_close(loc.grad, fx[f'{tag}.g_loc'], 5e-4, 5e-4, 'g_loc')
    assert_close(aw.grad, fx[f'{tag}.g_aw'], RT, AT, 'g_aw')


def test_msdeform_attn(golden):
    fx = golden('msdeform_attn')
    st = make_state(specs.msdeform(256, 3, 8, 4), 21, fx['wsum'])
    q, r, v = (T(fx[k]).requires_grad_() for k in ('query', 'refer', 'value'))
    out = O.msdeform_attn(q, r, v, fx['shapes'].tolist(), O.View(st), 8)
    assert_close(out, fx['out'], RT, AT, 'out')
    (out * T(fx['cot'])).sum().backward()
    for k, t in (('query', q), ('refer', r), ('value', v)):
        check_summary(fx, f'gin.{k}', t.grad, 5e-4, 5e-5)
    check_param_grads(fx, '', grads(st), 5e-4, 5e-5)


def test_msdeform_init_kat(golden):
    """KAT from the module's own init (transformer.py:234-250): ring-shaped offset bias."""
    fx = golden('msdeform_attn')
    th = torch.arange(8, dtype=torch.float32) * (2.0 * np.pi / 8)
    g = torch.stack([th.cos(), th.sin()], -1)
    g = (g / g.abs().max(-1, keepdim=True).values).view(8, 1, 1, 2).repeat(1, 3, 4, 1)
    g = g * torch.arange(1, 5, dtype=torch.float32).view(1, 1, 4, 1)
    assert_close(g.reshape(-1), fx['init_offsets_bias'], 1e-6, 1e-6)


def test_contrastive(golden):
    fx = golden('contrastive')
    st = make_state(specs.contrastive(), 31, fx['wsum'])
    x, w = T(fx['x']).requires_grad_(), T(fx['w']).requires_grad_()
    out = O.contrastive_head(x, w, O.View(st))
    assert_close(out, fx['out'], RT, AT)
    (out * T(fx['cot'])).sum().backward()
    check_summary(fx, 'gin.x', x.grad, RT, AT)
    check_summary(fx, 'gin.w', w.grad, RT, AT)
    check_param_grads(fx, '', grads(st), RT, 1e-4)
    assert abs(float(fx['init_bias'][0]) + 10.0) < 1e-6 and abs(float(fx['init_logit_scale']) - np.log(1 / 0.07)) < 1e-6


def test_decoder_layer(golden):
    fx = golden('decoder_layer')
    st = make_state(specs.decoder_layer(256, 8, 512, 3), 41, fx['wsum'])
    e, r, f, p = (T(fx[k]).requires_grad_() for k in ('embed', 'refer', 'feats', 'pos'))
    out = O.decoder_layer(e, r, f, fx['shapes'].tolist(), O.View(st), 8, T(fx['mask']), p)
    assert_close(out, fx['out'], RT, AT)
    (out * T(fx['cot'])).sum().backward()
    for k, t in (('embed', e), ('refer', r), ('feats', f), ('pos', p)):
        check_summary(fx, f'gin.{k}', t.grad, 5e-4, 5e-5)
    check_param_grads(fx, '', grads(st), 5e-4, 5e-5)
    with torch.no_grad():
        assert_close(O.decoder_layer(e, r, f, fx['shapes'].tolist(), O.View(st), 8, None, p), fx['out_nomask'], RT, AT)


def test_text_decoder(golden):
    fx = golden('text_decoder')
    spec = [(n.replace('decoder.layers', 'decoder.layers'), s, d) for n, s, d in specs.text_decoder_heads(256, 8, 512, 3)]
    st = make_state(spec, 42, fx['wsum'])
    e, r, f, t = (T(fx[k]).requires_grad_() for k in ('embed', 'refer', 'feats', 'text'))
    shapes = fx['shapes'].tolist()
    bb, sc = O.text_decoder(e, r, f, shapes, t, O.View(st), 8, 3, True, T(fx['mask']))
    assert_close(bb, fx['bboxes'], RT, AT, 'bboxes')
    assert_close(sc, fx['scores'], RT, 1e-4, 'scores')
    ((bb * T(fx['cot_b'])).sum() + (sc * T(fx['cot_s'])).sum()).backward()
    for k, x in (('embed', e), ('refer', r), ('feats', f), ('text', t)):
        check_summary(fx, f'gin.{k}', x.grad, 1e-3, 1e-4)
    check_param_grads(fx, '', grads(st), 1e-3, 1e-4)
    with torch.no_grad():
        bb, sc = O.text_decoder(e, r, f, shapes, t, O.View(st), 8, 3, False, None)
    assert_close(bb, fx['bboxes_eval'], RT, AT)
    assert_close(sc, fx['scores_eval'], RT, 1e-4)


def _targets(fx, pre=''):
    n_per = [int(v) for v in fx[pre + 'n_per']]
    return {'cls': T(fx[pre + 'cls']).long(), 'bboxes': T(fx[pre + 'bboxes']), 'batch_idx': T(fx[pre + 'batch_idx']).long(),
            'gt_groups': n_per}


@pytest.mark.parametrize('tag', ['A', 'B', 'C'])
def test_cdn_group(golden, tag):
    fx = golden('cdn')
    t = _targets(fx, tag + '.')
    nq, nd = [int(v) for v in fx[f'{tag}.cfg']]
    torch.manual_seed(1234)
    e, b, m, meta = O.cdn_group(t, 10, nq, T(fx[f'{tag}.class_embed']), nd, 0.5, 1.0, True)
    assert_close(e, fx[f'{tag}.dn_embed'], 1e-6, 1e-6)
    assert_close(b, fx[f'{tag}.dn_bbox'], 1e-5, 1e-5)
    assert torch.equal(m, T(fx[f'{tag}.mask']))
    assert meta['dn_num_group'] == int(fx[f'{tag}.num_group']) and meta['dn_num_split'] == fx[f'{tag}.split'].tolist()
    for i, p in enumerate(meta['dn_pos_idx']):
        assert torch.equal(p, T(fx[f'{tag}.pos_idx{i}']))


def test_cdn_group_off():
    t = {'cls': torch.zeros(0, dtype=torch.long), 'bboxes': torch.zeros(0, 4), 'batch_idx': torch.zeros(0, dtype=torch.long),
         'gt_groups': [0, 0]}
    assert O.cdn_group(t, 10, 5, torch.zeros(11, 4), 100, train=True) == (None, None, None, None)
    assert O.cdn_group(t, 10, 5, torch.zeros(11, 4), 100, train=False) == (None, None, None, None)


def test_riou(golden):
    fx = golden('riou')
    b1, b2 = T(fx['b1']).requires_grad_(), T(fx['b2']).requires_grad_()
    r = O.box_iou_xywh(b1, b2, riou=True)
    assert_close(r, fx['riou'], 1e-5, 1e-6)
    r.sum().backward()
    assert_close(b1.grad, fx['g_b1'], 1e-4, 1e-5)
    assert_close(b2.grad, fx['g_b2'], 1e-4, 1e-5)
    assert_close(O.box_iou_xywh(b1.detach(), b2.detach()), fx['iou'], 1e-5, 1e-6)


def test_matcher(golden):
    fx = golden('matcher')
    t = _targets(fx)
    idx = O.hungarian_match(T(fx['pred_bboxes']), T(fx['pred_scores']), t['bboxes'], t['cls'], t['gt_groups'])
    for i, (a, b) in enumerate(idx):
        assert torch.equal(a, T(fx[f'match{i}.src']).long()) and torch.equal(b, T(fx[f'match{i}.dst']).long())


def test_loss(golden):
    fx = golden('loss')
    t = _targets(fx)
    db, ds, eb, es = (T(fx[k]).requires_grad_() for k in ('dec_bboxes', 'dec_scores', 'enc_bboxes', 'enc_scores'))
    split = fx['split'].tolist()
    meta = {'dn_num_group': int(fx['num_group']), 'dn_num_split': split,
            'dn_pos_idx': [T(fx[f'pos_idx{i}']).long() for i in range(len(t['gt_groups']))]}
    dn_b, dec_b = torch.split(db, split, 2)
    dn_s, dec_s = torch.split(ds, split, 2)
    dec_b, dec_s = torch.cat([eb.unsqueeze(0), dec_b]), torch.cat([es.unsqueeze(0), dec_s])
    terms = O.rtdetr_loss(dec_b, dec_s, t, 10, dn_b, dn_s, meta)
    assert len(terms) == 12
    for k, v in terms.items():
        assert_close(v, fx[f'loss.{k}'], 1e-4, 1e-5, k)
    sum(terms.values()).backward()
    for k, x in (('dec_bboxes', db), ('dec_scores', ds), ('enc_bboxes', eb), ('enc_scores', es)):
        assert_close(x.grad, fx[f'g_{k}'], 5e-4, 1e-6, k)
    with torch.no_grad():
        for k, v in O.rtdetr_loss(dec_b, dec_s, t, 10).items():
            assert_close(v, fx[f'loss_nodn.{k}'], 1e-4, 1e-5, k)
        t0 = dict(t, cls=t['cls'][:0], bboxes=t['bboxes'][:0], batch_idx=t['batch_idx'][:0], gt_groups=[0, 0])
        for k, v in O.rtdetr_loss(dec_b, dec_s, t0, 10).items():
            assert_close(v, fx[f'loss_nogt.{k}'], 1e-4, 1e-5, k)


def _surrogate_scan(rec):
    def fn(u, delta, A, Bm, Cm, D, delta_bias, softplus=True):
        rec.update(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=delta_bias)
        Bn, KD, L = u.shape
        K = Bm.shape[1]
        dt = F.softplus(delta + delta_bias[None, :, None])
        bc = (Bm * Cm).sum(2)[:, :, None, :].expand(Bn, K, KD // K, L).reshape(Bn, KD, L)
        return u * D[None, :, None] + dt * bc * torch.exp(A.mean(1))[None, :, None]
    return fn


def test_vss_around_scan(golden):
    """Everything of VSSBlock except the S6 recurrence itself, pinned through forward_corev2's SelectiveScan= hook."""
    fx = golden('vss')
    x = T(fx['cs_x']).requires_grad_()
    xs = O.cross_scan(x)
    assert_close(xs, fx['cs_out'], 0, 0)
    (xs * T(fx['cs_cot'])).sum().backward()
    assert_close(x.grad, fx['cs_gx'], 1e-6, 1e-6)
    ys = T(fx['cm_ys']).requires_grad_()
    y = O.cross_merge(ys.flatten(3), 5, 7)
    assert_close(y, fx['cm_out'], 1e-6, 1e-6)
    (y * T(fx['cm_cot'])).sum().backward()
    assert_close(ys.grad, fx['cm_gys'], 1e-6, 1e-6)

    st = make_state(specs.vss_block(32), 51, fx['wsum'])
    rec = {}
    inp = T(fx['x']).requires_grad_()
    out = O.vss_block(inp, O.View(st), _surrogate_scan(rec))
    for k in ('u', 'delta', 'A', 'B', 'C', 'D', 'delta_bias'):
        assert_close(rec[k], fx[f'scan_in.{k}'], RT, AT, 'scan operand ' + k)
    assert_close(out, fx['out'], RT, AT)
    (out * T(fx['cot'])).sum().backward()
    check_summary(fx, 'gin.x', inp.grad, 5e-4, 5e-5)
    check_param_grads(fx, '', grads(st), 5e-4, 5e-5)


def _head_inputs(fx):
    ch, sizes = fx['ch'].tolist(), fx['sizes'].tolist()
    xs = [rnd((2, c, h, w), 70 + i).requires_grad_() for i, (c, (h, w)) in enumerate(zip(ch, sizes))]
    return xs, T(fx['text']).requires_grad_()


def test_meh_head(golden):
    fx = golden('head')
    hd, nq, nh, ndl, ffn, nc = [int(v) for v in fx['cfg']]
    st = make_state(specs.meh_head(nc, fx['ch'].tolist(), hd, nh, ndl, ffn, vss=False), 61, fx['wsum'])
    xs, text = _head_inputs(fx)
    t = _targets(fx)
    torch.manual_seed(4321)
    outs = O.meh_head(xs, text, t, O.View(st), nh, nq, ndl, nc, True, vss='identity')
    db, ds, eb, es, meta = outs
    assert meta['dn_num_split'] == fx['split'].tolist() and int(fx['n_invalid']) == 264
    for k, v in (('dec_bboxes', db), ('dec_scores', ds), ('enc_bboxes', eb), ('enc_scores', es)):
        assert_close(v, fx[k], 5e-4, 1e-4, k)
    sum((o * T(fx[f'cot{i}'])).sum() for i, o in enumerate((db, ds, eb, es))).backward()
    for i, x in enumerate(xs):
        check_summary(fx, f'gin.x{i}', x.grad, atol=1e-5, l2rel=5e-3)
    check_summary(fx, 'gin.text', text.grad, atol=1e-5, l2rel=5e-3)
    check_param_grads(fx, '', grads(st), atol=1e-5, l2rel=5e-3)
    for i in range(3):
        assert_close(st[f'input_proj.{i}.1.running_mean'], fx[f'bn{i}_mean'], RT, AT)
        assert_close(st[f'input_proj.{i}.1.running_var'], fx[f'bn{i}_var'], RT, AT)
    with torch.no_grad():
        y = O.meh_head(xs, text, None, O.View(st), nh, nq, ndl, nc, False, vss='identity')
    assert_close(y, fx['y_eval'], 5e-4, 1e-4, 'eval y')


def test_e2e(golden):
    """Full TAMTR graph, 256x256, VSS := identity (fixture flag), 12-term loss + sampled gradients + eval output."""
    fx = golden('e2e')
    assert int(fx['vss_identity']) == 1
    spec = specs.tamtr_model(10, vss=False)
    st = make_state(spec, int(fx['wseed']), fx['wsum'])
    n_par = sum(int(np.prod(s)) for n, s, d in spec if d.is_floating_point and not n.endswith(('running_mean', 'running_var')))
    vss_par = sum(int(np.prod(s)) for c in (128, 256, 512) for n, s, d in specs.vss_block(c))
    assert n_par + vss_par == 42124314
    assert int(fx['n_params']) == n_par
    S = int(fx['S'])
    batch = {'img': urnd((2, 3, S, S), 1), 'txt_feats': T(fx['txt']), 'cls': T(fx['cls']), 'bboxes': T(fx['bboxes']),
             'batch_idx': T(fx['batch_idx'])}
    torch.manual_seed(999)
    loss, items, terms = O.tamtr_loss(st, batch, True, vss='identity')
    assert_close(loss, fx['loss'], 1e-3, 1e-4, 'loss')
    assert_close(items, fx['loss_items'], 1e-3, 1e-4, 'items')
    loss.backward()
    gr = grads(st)
    none = sorted(k for k, g in gr.items() if g is None)
    assert none == sorted(fx['grad_none'].tolist()) and len(none) == 30
    check_param_grads(fx, '', gr, atol=1e-5, key='g', l2rel=1e-2)
    # the generator ran a second train-mode forward (same dn seed, no_grad) and stored its raw predictions
    t = _targets(fx)
    torch.manual_seed(999)
    with torch.no_grad():
        db, ds, eb, es, meta = O.tamtr_predict(st, batch['img'], batch['txt_feats'], t, True, vss='identity')
    assert meta['dn_num_split'] == fx['split'].tolist()
    for k, v in (('dec_bboxes', db), ('dec_scores', ds), ('enc_bboxes', eb), ('enc_scores', es)):
        assert_close(v, fx[k], 1e-3, 1e-4, k)
    assert_close(st['model.0.bn.running_mean'], fx['bn0_mean'], RT, AT)
    assert_close(st['model.16.attn.proj_conv.bn.running_mean'], fx['attn16_bn_mean'], RT, AT)
    with torch.no_grad():
        y = O.tamtr_predict(st, batch['img'], batch['txt_feats'], None, False, vss='identity')
    assert_close(y, fx['y_eval'], 1e-3, 1e-4, 'eval')


def test_selective_scan_properties():
    """a-9 scan is parity-unpinned: check the recurrence against a direct closed form and its limits."""
    torch.manual_seed(0)
    Bn, K, Dk, N, L = 2, 4, 3, 16, 9
    u, dl = torch.randn(Bn, K * Dk, L), torch.randn(Bn, K * Dk, L)
    A = -torch.exp(torch.randn(K * Dk, N) * 0.3)
    Bm, Cm = torch.randn(Bn, K, N, L), torch.randn(Bn, K, N, L)
    D, bias = torch.randn(K * Dk), torch.randn(K * Dk)
    y = O.selective_scan(u, dl, A, Bm, Cm, D, bias)
    dt = F.softplus(dl + bias[None, :, None]).double()
    Be, Ce = Bm.repeat_interleave(Dk, 1).double(), Cm.repeat_interleave(Dk, 1).double()
    want = torch.zeros(Bn, K * Dk, L, dtype=torch.float64)
    for t in range(L):  # y_t = sum_{s<=t} C_t . exp(A * sum_{r=s+1..t} dt_r) B_s dt_s u_s
        for s in range(t + 1):
            decay = torch.exp(A.double()[None] * dt[:, :, s + 1:t + 1].sum(-1, keepdim=True))
            want[:, :, t] += (Ce[..., t] * decay * Be[..., s]).sum(-1) * dt[:, :, s] * u[:, :, s].double()
    want += u.double() * D.double()[None, :, None]
    assert_close(y, want, 1e-4, 1e-5)
