#!/usr/bin/env python3
"""Validate a checkpoint on a YOLO-format dataset (the reference's valTAMTR.py flow): mAP50 / mAP50-95 / precision / recall.

    python tools/val.py --data dataset.yaml --text-feats clip_vitb32.npz --weights runs/train/TAMTR/best.pt [--split val]
"""
import argparse
import json
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', required=True)
    ap.add_argument('--text-feats', required=True)
    ap.add_argument('--weights', required=True)
    ap.add_argument('--split', default='val')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--imgsz', type=int, default=640)
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--conf', type=float, default=0.001)
    ap.add_argument('--iou', type=float, default=0.7)
    ap.add_argument('--raw', action='store_true', help='use the raw weights instead of the EMA copy')
    ap.add_argument('--no-fuse', action='store_true', help='keep BatchNorm layers separate (valTAMTR.py fuses, nn/autobackend.py:115)')
    args = ap.parse_args()

    import tamtr_amd  # noqa: F401
    from tamtr_amd import data as D, engine as E
    from tamtr_amd.model import RTDETRDetectionWorldModel
    dev = torch.device('cuda', 0)
    with open(args.data) as f:
        spec = yaml.safe_load(f)
    root = spec.get('path', os.path.dirname(os.path.abspath(args.data)))
    names = spec['names'] if isinstance(spec['names'], dict) else dict(enumerate(spec['names']))
    tf = D.TextFeatures.load(args.text_feats)
    ds = D.PromptDetDataset(os.path.normpath(os.path.join(root, spec[args.split])), names, args.imgsz, augment=False)
    loader = D.build_dataloader(ds, args.batch, args.workers, shuffle=False)
    model = RTDETRDetectionWorldModel(nc=len(names)).to(dev)
    ck = torch.load(args.weights, map_location=dev)
    model.load_state_dict(ck['model' if args.raw else 'ema'])
    model.set_text_features(tf.encode([v.split('/')[0] for v in names.values()])[None].to(dev))
    model.eval()
    model.autocast_dtype = torch.bfloat16 if args.dtype == 'bf16' else None    # predict() opens its own autocast region from this
    if not args.no_fuse:
        model.fuse()
    res = E.validate(model, (D.preprocess_batch(b, None, dev) for b in loader), imgsz=args.imgsz, conf=args.conf, iou=args.iou,
                     autocast_dtype=torch.bfloat16 if args.dtype == 'bf16' else None)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
