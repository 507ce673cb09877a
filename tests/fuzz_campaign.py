#!/usr/bin/env python3
"""Time-bounded fuzz campaign (not collected by pytest): the seeded generators of test_gpu_fuzz.py over many more
seeds and larger candidate counts (single-wave, single-launch and multi-kernel NMS paths), every result compared
bit for bit with the CPU oracle.  Prints one progress line per 25 cases and a JSON summary.

    python tests/fuzz_campaign.py [seconds=240] [first_seed=1000]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vortex_amd as V  # noqa: E402
from vortex_amd import ops  # noqa: E402
from oracle import vortex_oracle as O  # noqa: E402
from tests.test_gpu_fuzz import _boxes  # noqa: E402

DEV = torch.device("cuda:0")


def nms_case(seed):
    g = torch.Generator().manual_seed(50_000 + seed)
    hi = [64, 300, 3000, 17000, 40000][seed % 5]          # wave path / pair path / LDS path / edge of LDS / global sort
    K = int(torch.randint(1, hi + 1, (1,), generator=g))
    C = [1, 2, 5, 21, 80, 1000, 4096][seed % 7]
    kind = ["normal", "grid", "dup", "inverted"][(seed // 5) % 4]
    if kind in ("dup", "grid") and K > 6000:
        K = 6000                                          # O(K^2) oracle on heavily overlapping boxes
    boxes = _boxes(kind, K, g).contiguous()
    scores = torch.rand(K, generator=g)
    if seed % 3 == 0:
        scores = (scores * 16).floor() / 16
    cls = torch.randint(0, C, (K,), generator=g)
    thr = [0.0, 0.1, 0.25, 0.3333333, 0.45, 0.5, 0.75, 1.0, 0.2, 0.6][seed % 10]
    ref = O.batched_nms(boxes, scores, cls, torch.tensor([thr]))
    got = V.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), torch.tensor([thr])).cpu()
    return torch.equal(got, ref), dict(kind="nms", seed=seed, K=K, C=C, box=kind, thr=thr)


def yolo_case(seed):
    g = torch.Generator().manual_seed(60_000 + seed)
    S = 32 * int(torch.randint(1, 11, (1,), generator=g))
    C = [1, 2, 3, 7, 20, 80, 91][seed % 7]
    B = 1 + seed % 4
    sthr = [0.06, 0.5, 0.9, 0.0, 0.3, 0.97][seed % 6]
    ithr = [0.2, 0.45, 1.0, 0.0, 0.6][seed % 5]
    heads = O.synth_yolo_heads(B, S, C, seed=70_000 + seed)
    pred = O.yolo_head_eval(heads, S)
    st, it = torch.tensor([sthr]), torch.tensor([ithr])
    # bit-exact boundary: decoded head -> decoder -> NMS
    ref = O.yolo_postprocess(pred, st, it, S)
    got = V.YoloV3PostProcess(S)(pred.to(DEV), st, it)
    ok = len(got) == B and all(a.shape == e.shape and torch.equal(a.cpu(), e) for a, e in zip(got, ref))
    # raw-head boundary: the three decode modes agree bit for bit, and the one-call pipeline equals NMS on them
    dheads = [h.to(DEV) for h in heads]
    a = ops.yolo_fused_decode(dheads, S, sthr)
    t = ops.yolo_fused_decode(dheads, S, sthr, threshold_first=True)
    ok = ok and torch.equal(a.meta, t.meta)
    res = V.YoloV3FusedPostProcess(S)(dheads, st, it)
    for b, k in enumerate(a.counts.cpu().tolist()):
        ok = ok and torch.equal(a.det[b, :k], t.det[b, :k]) and torch.equal(a.scores[b, :k], t.scores[b, :k])
        keep = O.batched_nms(a.boxes[b, :k].cpu(), a.scores[b, :k].cpu(), a.cls[b, :k].cpu(), it)
        ok = ok and torch.equal(res[b].cpu(), a.det[b, :k].cpu()[keep])
    # the one-launch head decode (three scales + concat) against the oracle's YoloV3Layer restatement (1e-5)
    dec = V.YoloV3HeadDecode(S, C)(*dheads).cpu()
    ok = ok and dec.shape == pred.shape and bool(((dec - pred).abs() <= 1e-5 * pred.abs() + 1e-6).all())
    return ok, dict(kind="yolo", seed=seed, S=S, C=C, B=B, sthr=sthr, ithr=ithr)


def prior_case(seed):
    g = torch.Generator().manual_seed(80_000 + seed)
    P = int(torch.randint(1, 3000, (1,), generator=g))
    Lm = [0, 5, 0, 2][seed % 4]
    NL = [2, 2, 21, 4][seed % 4]
    B = 1 + seed % 3
    priors = torch.rand(P, 4, generator=g) * 0.5 + 0.05
    x = O.synth_prior_head(B, P, Lm, NL, seed=90_000 + seed)
    sthr = [0.5, 0.05, 0.9, 0.3][seed % 4]
    st, it = torch.tensor([sthr]), torch.tensor([0.4])
    pp = V.PriorBoxPostProcess(priors.to(DEV), n_landmarks=Lm).to(DEV)
    got = pp(x.to(DEV), st, it)
    # tolerance-class decode: compare the NMS on the GPU's own candidates with the oracle's NMS (bit-exact part)
    c = pp.decoder.decode_batch(x.to(DEV), st) if hasattr(pp.decoder, "decode_batch") else None
    ok = len(got) == B
    if c is not None:
        from tests.test_gpu_parity import close, same
        for b, k in enumerate(c.counts.cpu().tolist()):
            keep = O.batched_nms(c.boxes[b, :k].cpu(), c.scores[b, :k].cpu(), c.cls[b, :k].cpu(), it)
            ok = ok and torch.equal(got[b].cpu(), c.det[b, :k].cpu()[keep])
            # decode itself against the oracle (tolerance class: exp / softmax, 1e-5 rel)
            bb, sc, ci, det = O.prior_box_decode(x[b:b + 1], priors, st, Lm)
            if k != bb.shape[0]:                         # a confidence within the tolerance of the threshold
                n_hi = O.prior_box_decode(x[b:b + 1], priors, st * (1 - 3e-5), Lm)[0].shape[0]
                n_lo = O.prior_box_decode(x[b:b + 1], priors, st * (1 + 3e-5), Lm)[0].shape[0]
                ok = ok and n_lo <= k <= n_hi
                continue
            try:
                close(c.boxes[b, :k], bb); close(c.scores[b, :k], sc)
                cols = [0, 1, 2, 3, 4] + list(range(6, 6 + 2 * Lm))
                close(c.det[b, :k][:, cols], det[0][:, cols])
                amb = torch.softmax(x[b, :, 4 + 2 * Lm:], 1)[:, 1:].topk(min(2, NL - 1), 1).values
                if NL > 2 and bool(((amb[:, 0] - amb[:, 1]) <= 2e-5 * amb[:, 0]).any()):
                    pass                                 # two classes within the tolerance: label may differ
                else:
                    same(c.cls[b, :k], ci)
            except AssertionError:
                ok = False
    return ok, dict(kind="prior", seed=seed, P=P, L=Lm, NL=NL, B=B, sthr=sthr)


def ragged_case(seed):
    """B images, counts from 0 to beyond the single-launch limit in ONE call (both NMS paths side by side), top_k."""
    g = torch.Generator().manual_seed(110_000 + seed)
    B = 2 + seed % 4
    cap = [900, 5000, 17500, 24000][seed % 4]
    C = [1, 3, 80, 4096, 20, 256, 4][(seed // 4) % 7]
    counts = torch.randint(0, cap + 1, (B,), generator=g)
    counts[0] = cap
    if B > 2:
        counts[1] = 0
    boxes = torch.stack([_boxes("normal", cap, g) for _ in range(B)]).contiguous()
    scores = torch.rand(B, cap, generator=g)
    cls = torch.randint(0, C, (B, cap), generator=g)
    det = torch.randn(B, cap, 6, generator=g)
    top_k = [None, 100, 1, 3000, 2048][seed % 5]
    thr = [0.3, 0.5, 0.7][seed % 3]
    if seed % 11 == 0:
        scores = (scores * 4).floor() / 4                # heavy ties: the top_k select descends into the index bits
    if seed % 13 == 0:
        cls[0, : cap * 3 // 4] = C - 1                    # one long segment (beyond the LDS sort when cap > 21845)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(
        boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), counts.to(torch.int32).to(DEV), thr, C, top_k)
    ok = ops.status_ok(int(status.item()))
    kc = keep_counts.cpu().tolist()
    for b in range(B):
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k is not None:
            ref = ref[:top_k]
        ok = ok and kc[b] == ref.numel() and torch.equal(keep_idx[b, :kc[b]].cpu(), ref) \
            and torch.equal(out_det[b, :kc[b]].cpu(), det[b, :n][ref])
    return ok, dict(kind="ragged", seed=seed, B=B, cap=cap, C=C, top_k=top_k, thr=thr)


def pre_case(seed):
    """Normalize / FlipNormalize on arbitrary H, W (odd sizes, unaligned planes) and the u8 resize + fused
    resize-normalize on arbitrary source / destination sizes, stretch and letterbox: bit-exact."""
    import numpy as np
    g = torch.Generator().manual_seed(120_000 + seed)
    B = 1 + seed % 3
    H, W = (int(v) for v in torch.randint(1, 200, (2,), generator=g))
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    mean = (torch.rand(3, generator=g)).tolist()
    std = (torch.rand(3, generator=g) * 0.9 + 0.1).tolist()
    flip = bool(seed & 1)
    mod = (V.FlipNormalize if flip else V.Normalize)(mean, std).to(DEV)
    ok = torch.equal(mod(img.to(DEV)).cpu(), O.normalize(img, mean, std, flip=flip))
    oh, ow = (int(v) for v in torch.randint(1, 160, (2,), generator=g))
    lb = bool(seed & 2)
    ref = np.stack([O.resize_bilinear_u8(im, oh, ow, lb) for im in img.numpy()])
    y = ops.resize_u8(img.to(DEV), (oh, ow), letterbox=lb).cpu().numpy()
    ok = ok and np.array_equal(y, ref)
    z = ops.preprocess(img.to(DEV), mean, std, 255, out_size=(oh, ow), letterbox=lb)
    ok = ok and torch.equal(z.cpu(), O.normalize(torch.from_numpy(ref), mean, std))
    return ok, dict(kind="pre", seed=seed, B=B, H=H, W=W, oh=oh, ow=ow, letterbox=lb, flip=flip)


def layer_case(seed):
    """Raw-head boundary at any S up to 640 (R > 16384 from S = 544: optimistic NMS, deferred + redo when dense):
    fused == threshold-first == YoloV3Layer x3 + decoder, one-call pipeline with top_k == oracle NMS on them."""
    g = torch.Generator().manual_seed(130_000 + seed)
    S = 32 * int(torch.randint(1, 21, (1,), generator=g))
    C = [1, 2, 5, 20, 80, 243][seed % 6]
    B = 1 + seed % 2 if S > 416 else 1 + seed % 5
    thr = [0.9, 0.5, 0.01, 0.97, 0.0, 0.7][(seed // 2) % 6]
    top_k = [None, 10, 300][seed % 3]
    heads = [h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=140_000 + seed)]
    a = ops.yolo_fused_decode(heads, S, thr)
    t = ops.yolo_fused_decode(heads, S, thr, threshold_first=True)
    out = torch.empty((B, ops.yolo_rows(S), 5 + C), device=DEV)
    off = 0
    for h, mask in zip(heads, O.ANCHOR_MASKS):
        G = h.shape[-1]
        V.YoloV3Layer(S, (G, G), list(mask), O.COCO_ANCHORS, C).to(DEV).eval()(h, out=out, row_offset=off)
        off += 3 * G * G
    u = ops.yolo_decode(out, S, thr)
    ok = torch.equal(a.meta, t.meta) and torch.equal(a.meta, u.meta)
    it = torch.tensor([[0.2, 0.45, 0.6][seed % 3]])
    res = V.YoloV3FusedPostProcess(S, top_k=top_k)(heads, torch.tensor([thr]), it)
    for b, k in enumerate(a.counts.cpu().tolist()):
        for x, y, z in ((a.boxes, t.boxes, u.boxes), (a.scores, t.scores, u.scores), (a.cls, t.cls, u.cls), (a.det, t.det, u.det)):
            ok = ok and torch.equal(x[b, :k], y[b, :k]) and torch.equal(x[b, :k], z[b, :k])
        if k <= 12000:                                   # O(K^2) oracle
            keep = O.batched_nms(a.boxes[b, :k].cpu(), a.scores[b, :k].cpu(), a.cls[b, :k].cpu(), it)
            if top_k is not None:
                keep = keep[:top_k]
            ok = ok and torch.equal(res[b].cpu(), a.det[b, :k].cpu()[keep])
    return ok, dict(kind="layer", seed=seed, S=S, C=C, B=B, thr=thr, top_k=top_k)


def format_case(seed):
    import numpy as np
    g = torch.Generator().manual_seed(150_000 + seed)
    B, cap, D = 1 + seed % 7, int(torch.randint(1, 400, (1,), generator=g)), [6, 16, 7, 64][seed % 4]
    det = torch.randn(B, cap, D, generator=g)
    counts = torch.randint(0, cap + 1, (B,), generator=g).to(torch.int32)
    ncol = int(torch.randint(1, 65, (1,), generator=g))
    cols = torch.randint(0, D, (ncol,), generator=g).tolist()
    kinds = torch.randint(0, 3, (ncol,), generator=g).tolist()
    wh = (torch.rand(B, 2, generator=g) * 2000 + 1).float()
    rows = ops.format_results(det.to(DEV), counts.to(DEV), cols, kinds, wh.to(DEV)).cpu()
    exp = []
    for b in range(B):
        e = det[b, :int(counts[b])][:, cols].clone()
        for j, kd in enumerate(kinds):
            if kd:
                e[:, j] = e[:, j] * wh[b, kd - 1]
        exp.append(e)
    exp = torch.cat(exp) if exp else torch.zeros(0, ncol)
    ok = torch.equal(rows[:exp.shape[0]], exp)
    packed = ops.pack_rows(det.to(DEV), counts.to(DEV)).cpu()
    flat = torch.cat([det[b, :int(counts[b])] for b in range(B)])
    ok = ok and torch.equal(packed[:flat.shape[0]], flat)
    return ok, dict(kind="format", seed=seed, B=B, cap=cap, D=D, ncol=ncol)


def detr_case(seed):
    from tests.test_gpu_parity import _detr_check
    g = torch.Generator().manual_seed(160_000 + seed)
    B, Q = 1 + seed % 9, int(torch.randint(1, 700, (1,), generator=g))
    NL = [2, 3, 92, 65, 1204][seed % 5]
    thr = [0.5, 0.3, 0.0, 0.9, 0.05][(seed // 5) % 5]
    x = O.synth_detr_outputs(B, Q, NL, 170_000 + seed)
    try:
        _detr_check(x, thr, O.detr_postprocess(x, torch.tensor([thr])))
        ok = True
    except AssertionError:
        ok = False
    lg = torch.randn(B * 3, NL, generator=g) * 3
    r, e = V.SoftmaxPostProcess()(lg.to(DEV)).cpu(), O.softmax_postprocess(lg)
    p = torch.softmax(lg, 1).topk(2, 1).values
    clear = (p[:, 0] - p[:, 1]) > 2e-5 * p[:, 0]
    ok = ok and torch.equal(r[clear, 0], e[clear, 0]) and torch.allclose(r[:, 1], e[:, 1], rtol=1e-5, atol=0)
    return ok, dict(kind="detr", seed=seed, B=B, Q=Q, NL=NL, thr=thr)


def nan_case(seed):
    """One NaN anywhere in the input -> ValueError (postprocess/yolov3.py:35-36) on every YOLO entry; without
    it no error."""
    g = torch.Generator().manual_seed(180_000 + seed)
    S, C, B = 32 * (1 + seed % 6), [1, 3, 80][seed % 3], 1 + seed % 3
    heads = O.synth_yolo_heads(B, S, C, seed=190_000 + seed)
    hi = int(torch.randint(0, 3, (1,), generator=g))
    flat = heads[hi].view(-1)
    pos = int(torch.randint(0, flat.numel(), (1,), generator=g))
    st, it = torch.tensor([[0.9, 0.3][seed % 2]]), torch.tensor([0.45])
    pred = O.yolo_head_eval(heads, S)
    ok = True
    try:
        V.YoloV3FusedPostProcess(S)([h.to(DEV) for h in heads], st, it)
        V.YoloV3PostProcess(S)(pred.to(DEV), st, it)
    except ValueError:
        ok = False
    flat[pos] = float("nan")
    pred2 = pred.clone()
    pred2.view(-1)[int(torch.randint(0, pred2.numel(), (1,), generator=g))] = float("nan")
    for fn in (lambda: V.YoloV3FusedPostProcess(S)([h.to(DEV) for h in heads], st, it),
               lambda: V.YoloV3PostProcess(S)(pred2.to(DEV), st, it)):
        try:
            fn()
            ok = False
        except ValueError:
            pass
    return ok, dict(kind="nan", seed=seed, S=S, C=C, B=B, head=hi, pos=pos)


def declared_case(seed):
    """ops.batched_nms with the class count declared: batches of 1-3 images of different fill, capacities on both sides
    of the single-launch kernel (so the class-partitioned path, the ordered kept runs + merge rounds, the top-k select,
    the general path for an oversized class and the optimistic call's deferral all come up)."""
    g = torch.Generator().manual_seed(90_000 + seed)
    K = int(torch.randint(200, [3000, 20000, 45000][seed % 3], (1,), generator=g))
    C = [4, 7, 20, 80, 256][seed % 5]
    kind = ["normal", "grid", "dup", "inverted"][(seed // 3) % 4]
    if kind in ("dup", "grid"):
        K = min(K, 9000)
    B = 1 + seed % 3
    top_k = [None, 100, 2048, 2049, 6000][(seed // 2) % 5]
    thr = [0.2, 0.45, 0.5, 0.7][seed % 4]
    optimistic = K > ops.NMS_SINGLE_LAUNCH_MAX and seed % 2 == 0
    boxes = torch.stack([_boxes(kind, K, g) for _ in range(B)]).contiguous()
    scores = torch.rand(B, K, generator=g)
    if seed % 4 == 1:
        scores = (scores * 32).floor() / 32
    cls = torch.randint(0, C, (B, K), generator=g)
    if seed % 7 == 0:
        cls[0, : int(K * 0.7)] = C - 1
    counts = torch.tensor([K if b == 0 else int(torch.randint(0, K + 1, (1,), generator=g)) for b in range(B)], dtype=torch.int32)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV),
                                                             counts.to(DEV), thr, n_classes=C, top_k=top_k, optimistic=optimistic)
    st = int(status.item())
    deferred = bool(st & V._lib.VX_STATUS_DEFERRED)
    ok = ops.status_ok(st & ~V._lib.VX_STATUS_DEFERRED)
    if deferred:                                          # only an image with one class beyond 16384 candidates may be deferred
        ok = ok and optimistic and any(int(torch.bincount(cls[b, :int(counts[b])], minlength=C).max()) > 16384 for b in range(B))
        keep_idx, out_det, keep_counts, status = ops.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV),
                                                                 counts.to(DEV), thr, n_classes=C, top_k=top_k)
        ok = ok and ops.status_ok(int(status.item()))
    for b in range(B):
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k:
            ref = ref[:top_k]
        kc = int(keep_counts[b].item())
        ok = ok and kc == ref.numel() and torch.equal(keep_idx[b, :kc].cpu(), ref) and torch.equal(out_det[b, :kc].cpu(), det[b, ref])
    return ok, dict(kind="declared", seed=seed, K=K, C=C, B=B, box=kind, thr=thr, top_k=top_k, optimistic=optimistic)


def bitmask_case(seed):
    """capacities beyond the single-launch kernel: whole-chip bitmask NMS of large class segments (image route and
    class partition), its fall-backs, the score-prefix path of top_k and the optimistic call (tests/test_gpu_bitmask.py
    with random shapes)."""
    from tests.test_gpu_bitmask import make
    g = torch.Generator().manual_seed(110_000 + seed)
    K = int(torch.randint(16385, [24000, 52000, 100001][seed % 3], (1,), generator=g))
    C = [1, 1, 3, 20, 80][seed % 5]
    B = 1 + seed % 3
    lo = [0.01, 0.02, 0.05, 0.2][(seed // 3) % 4]
    wh = (lo, lo * [3, 10, 2][(seed // 5) % 3])
    fill = [1.0, 0.4, 0.1][(seed // 2) % 3]                # image route needs <= 16384 candidates per image
    n = [max(1, int(K * fill * float(torch.rand(1, generator=g)))) if (C <= 3 or b) else K for b in range(B)]
    if C <= 3:
        n = [min(v, 16000) for v in n]
    kw = dict(dup=int(n[0] * 0.8) if seed % 11 == 0 else 0, untame=seed % 13 == 0, ties=[0, 0, 7, 1000][seed % 4],
              pixel=seed % 6 == 0)
    top_k = [None, None, 100, 300, 5000][(seed // 4) % 5]
    thr = [0.45, 0.3, 0.6, 0.2][seed % 4]
    optimistic = seed % 2 == 1
    boxes, scores, cls, det, counts = make(B, K, C, seed=120_000 + seed, n=n, wh=wh, **kw)
    d = [t.to(DEV) if t is not None else None for t in (boxes, scores, cls, det, counts)]
    keep_idx, out_det, keep_counts, status = ops.batched_nms(*d, thr, C, top_k, optimistic=optimistic)
    st = int(status.item())
    if st & V._lib.VX_STATUS_DEFERRED:
        keep_idx, out_det, keep_counts, status = ops.batched_nms(*d, thr, C, top_k)
        st = int(status.item()) | (0 if optimistic else 1 << 30)      # a deferral without the optimistic flag is a failure
    ok = ops.status_ok(st)
    kc = keep_counts.cpu().tolist()
    for b in range(B):
        ref = O.batched_nms(boxes[b, :n[b]], scores[b, :n[b]], cls[b, :n[b]], torch.tensor([thr]))
        if top_k:
            ref = ref[:top_k]
        ok = ok and kc[b] == ref.numel() and torch.equal(keep_idx[b, :kc[b]].cpu(), ref) \
            and torch.equal(out_det[b, :kc[b]].cpu(), det[b][ref])
    return ok, dict(kind="bitmask", seed=seed, K=K, C=C, B=B, n=n, wh=wh, thr=thr, top_k=top_k, optimistic=optimistic, **kw)


CASES = (nms_case, ragged_case, yolo_case, prior_case, pre_case, layer_case, format_case, detr_case, nan_case,
         nms_case, layer_case, declared_case, declared_case, bitmask_case, bitmask_case)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, n, fails = time.time(), 0, []
    per_kind = {}
    seed = first
    while time.time() - t0 < budget:
        fn = CASES[seed % len(CASES)]
        ok, info = fn(seed)
        per_kind[info["kind"]] = per_kind.get(info["kind"], 0) + 1
        n += 1
        if not ok:
            fails.append(info)
            print("FAIL", json.dumps(info), flush=True)
        if n % 25 == 0:
            print("cases %d  failures %d  %.0f s" % (n, len(fails), time.time() - t0), flush=True)
        seed += 1
    print(json.dumps({"cases": n, "per_kind": per_kind, "n_failures": len(fails), "failures": fails[:12], "seconds": round(time.time() - t0, 1),
                      "first_seed": first}), flush=True)
    sys.exit(1 if fails else 0)
