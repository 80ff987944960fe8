#!/usr/bin/env python3
"""Where does the bf16 mode's error against the fp32 oracle come from?  

The whole graph at 640 x 640 (2 images: the CPU oracle with the C scan twin takes ~15 s), the oracle's discrete choices injected
(top-k picks, Hungarian pairs: head.fixed_topk / criterion.fixed_matches), MIOpen on its deterministic solvers (TAMTR_DETERMINISTIC=1:
NCHW trunk, no atomic split-K sums), so that every figure is ARITHMETIC and reproducible.  The forward is run stage by stage with bf16
autocast switched on for a chosen set of stages and fp32 everywhere else:

    trunk      model.model[:-1]            (GELAN + BTA-PAN: MIOpen convolutions, BatchNorm + SiLU kernels, gates)
    vss        head.VSSBlocks              (in_proj / out_proj / MLP GEMMs in bf16; the scan itself is fp32 in both modes)
    proj       head.input_proj             (1x1 projection GEMM + BatchNorm -> the token memory)
    enc        head._get_decoder_input     (enc_output GEMM + LayerNorm, enc_score_head, top-k gather, enc_bbox_head)
    decoder    head.decoder + heads        (self-attention, value_proj GEMM, deformable gather, FFN, bbox heads, contrastive head)

Configurations: fp32 everywhere; each stage alone; the pipeline switched on cumulatively front to back; everything (= the benchmarked mode).
Per configuration, against the fp32 CPU oracle: relative error of the loss and of the worst of the 12 terms, box error (sigmoid space),
class-logit error (scale ~10), encoder-score error.  One JSON object on stdout (and --out).

    TAMTR_DETERMINISTIC=1 python3 tools/bf16_attribution.py --out profiles/r04_bf16_attribution.json
"""
import argparse, json, os, sys, time
os.environ.setdefault('TAMTR_DETERMINISTIC', '1')
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import selscan_c, tamtr_oracle as O          # the checker (tools/ may use it: this is a parity measurement)
from weights import fill_state
from test_gpu_fullsize import _bench_batch
from tamtr_amd import tuning
from tamtr_amd.model import RTDETRDetectionWorldModel

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
ap.add_argument('--imgsz', type=int, default=640)
args = ap.parse_args()
note = tuning.use_tuned_convolutions('shipped')    # TAMTR_DETERMINISTIC=1 -> deterministic solvers
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10)
for m in model.modules():
    if hasattr(m, 'drop_prob'):
        m.drop_prob = 0.0
state = fill_state(model.state_dict(), 83)
model.load_state_dict(state)
model.cuda().train()
assert not model.channels_last, 'deterministic mode runs the NCHW trunk'
batch = _bench_batch(2, args.imgsz, 1)
bidx = batch['batch_idx']
tg_host = {'cls': batch['cls'], 'bboxes': batch['bboxes'], 'batch_idx': bidx, 'gt_groups': [int((bidx == i).sum()) for i in range(2)]}
t0 = time.time()
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
so = {k: v.clone() for k, v in state.items()}
with torch.no_grad():
    torch.manual_seed(5)
    O.TRACE = trace = {}
    try:
        lref, iref, tref = O.tamtr_loss(so, batch, True, scan_fn=selscan_c.scan)
    finally:
        O.TRACE = None
    torch.manual_seed(5)
    rdb, rds, reb, res_, rmeta = O.tamtr_predict(so, batch['img'], batch['txt_feats'], tg_host, True, scan_fn=selscan_c.scan)
mt = trace['matches']
choices = {'top': trace['top'][0], 'matches': [mt[1], mt[2], mt[3], mt[0]]}
print(f'[attr] oracle: {time.time() - t0:.1f} s, loss {float(lref):.5f}', file=sys.stderr, flush=True)
from staged import HIP_PATH, STAGES, errors, staged_forward
ref = (float(lref), {k: float(v) for k, v in tref.items()}, rdb, rds, reb, res_)


def staged(on):
    r = errors(staged_forward(model, state, batch, tg_host, choices, on), ref)
    r['bf16_stages'] = sorted(on, key=STAGES.index)
    return r


rows = []
from staged import HIP_PATH as _HP, STAGES as _ST
configs = [[]] + [[s] for s in _ST] + [list(_HP)] + [_ST[:k] for k in range(2, len(_ST) + 1)]
for on in configs:
    r = staged(set(on))
    r2 = staged(set(on))
    r['reproducible'] = r2['loss'] == r['loss'] and r2['cls_logit_abs_max'] == r['cls_logit_abs_max']
    rows.append(r)
    print(f"[attr] bf16 in {'+'.join(on) or '(nothing: fp32)':32s} loss_rel {r['loss_rel']:.2e}  worst term {r['term_rel_max']:.2e} ({r['worst_term']})  box max {r['box_abs_max']:.2e} "
          f"mean {r['box_abs_mean']:.2e}  cls logit max {r['cls_logit_abs_max']:.3f} mean {r['cls_logit_abs_mean']:.3f}  enc score max {r['enc_score_abs_max']:.3f}  "
          f"reproducible {r['reproducible']}", file=sys.stderr, flush=True)
out = {'imgsz': args.imgsz, 'batch': 2, 'convolutions': note, 'oracle_loss': float(lref), 'logit_scale': float(rds.abs().mean()),
       'what': 'bf16 autocast enabled stage by stage, fp32 elsewhere; oracle top-k picks and Hungarian pairs injected; errors against the fp32 CPU oracle',
       'rows': rows}
print(json.dumps(out))
if args.out:
    json.dump(out, open(args.out, 'w'), indent=1)
