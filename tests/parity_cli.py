#!/usr/bin/env python3
"""Parity CLI: the repo's counterpart of the reference's export check (scripts/export_check/onnx_check.py:249-314:
random uint8 input, thresholds 0.0 / 1.0 by default, predictor output of one backend compared with another with
isclose(rtol 1e-3, atol 1e-6)).  Here the two "backends" are the HIP path (through the C ABI) and the CPU oracle,
on the same seeded synthetic inputs (test infrastructure: it lives under tests/ because it runs the oracle as the
checker), and the bars are the repo's own: bit-exact for integer / index work,
1e-5 relative for the sigmoid/exp/softmax stages.  One line per check, exit code 1 on any failure.

    python tests/parity_cli.py [--score-threshold 0.0 --iou-threshold 1.0] [--batch 4] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vortex_amd as V  # noqa: E402
from vortex_amd import ops  # noqa: E402
from oracle import vortex_oracle as O  # noqa: E402

RTOL, ATOL = 1e-5, 1e-6
FAILED = []


def report(name, ok, detail=""):
    print("%-58s %s %s" % (name, "ok  " if ok else "FAIL", detail), flush=True)
    if not ok:
        FAILED.append(name)


def exact(a, b):
    a, b = a.detach().cpu().numpy(), (b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b))
    return a.shape == b.shape and np.array_equal(a, b)


def close(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return a.shape == b.shape and bool(((a - b).abs() <= ATOL + RTOL * b.abs()).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--score-threshold", type=float, default=0.0)     # onnx_check.py:282-285
    ap.add_argument("--iou-threshold", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--img-size", type=int, default=416)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    S, C, B = args.img_size, 80, args.batch
    st, it = torch.tensor([args.score_threshold]), torch.tensor([args.iou_threshold])

    img = O.synth_images(B, S, seed=args.seed)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    report("Normalize u8 NHWC -> f32 NCHW (bit-exact)", exact(V.Normalize(mean, std).to(dev)(img.to(dev)), O.normalize(img, mean, std)))

    heads = O.synth_yolo_heads(B, S, C, seed=args.seed)
    pred = O.yolo_head_eval(heads, S)
    out = torch.empty((B, ops.yolo_rows(S), 5 + C), device=dev)
    off = 0
    for h, mask in zip(heads, O.ANCHOR_MASKS):
        G = h.shape[-1]
        V.YoloV3Layer(S, (G, G), list(mask), O.COCO_ANCHORS, C).to(dev).eval()(h.to(dev), out=out, row_offset=off)
        off += 3 * G * G
    report("YoloV3Layer x3 + concat (1e-5 rel)", close(out, pred))

    ref = O.yolo_postprocess(pred, st, it, S)
    got = V.YoloV3PostProcess(S)(pred.to(dev), st, it)
    report("YoloV3PostProcess on the oracle's decoded head (bit-exact)",
           len(got) == len(ref) and all(exact(g, r) for g, r in zip(got, ref)),
           "detections/image %s" % [int(r.shape[0]) for r in ref])

    fused = V.YoloV3FusedPostProcess(S)([h.to(dev) for h in heads], st, it)
    unf = V.YoloV3PostProcess(S)(out, st, it)
    report("fused raw-head path == layer-by-layer path on the GPU (bit-exact)", all(exact(f, u) for f, u in zip(fused, unf)))

    boxes, scores, cls = O.synth_stress_boxes(20000, 80, args.seed)
    report("batched_nms 20000 boxes x 80 classes, iou 0.45 (bit-exact)",
           exact(V.batched_nms(boxes.to(dev), scores.to(dev), cls.to(dev), torch.tensor([0.45])),
                 O.batched_nms(boxes, scores, cls, torch.tensor([0.45]))))

    priors = O.make_priors(640)
    x = O.synth_prior_head(2, priors.shape[0], 5, 2, seed=args.seed)
    sthr = max(args.score_threshold, 0.02)
    c = ops.prior_decode(x.to(dev), priors.to(dev), 5, sthr)
    ok = True
    for b, k in enumerate(c.counts.cpu().tolist()):
        bb, sc, ci, det = O.prior_box_decode(x[b:b + 1], priors, torch.tensor([sthr]), 5)
        ok = ok and k == bb.shape[0] and close(c.boxes[b, :k], bb) and close(c.scores[b, :k], sc) and exact(c.cls[b, :k], ci)
    report("prior-box decode + landmarks, RetinaFace-640 (1e-5 rel, labels exact)", ok)

    xd = O.synth_detr_outputs(B, 100, 92, args.seed)
    dthr = min(max(args.score_threshold, 0.05), 0.95)
    gd, rd = V.DETRPostProcess()(xd.to(dev), score_threshold=torch.tensor([dthr])), O.detr_postprocess(xd, torch.tensor([dthr]))
    report("DETRPostProcess (boxes/labels exact, conf 1e-5 rel)",
           all(g.shape == r.shape and exact(g[:, :4], r[:, :4]) and exact(g[:, 5], r[:, 5]) and close(g[:, 4], r[:, 4])
               for g, r in zip(gd, rd)))

    print("%d checks, %d failed" % (7, len(FAILED)))
    return 1 if FAILED else 0


if __name__ == "__main__":
    sys.exit(main())
