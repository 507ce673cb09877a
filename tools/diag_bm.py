#!/usr/bin/env python3
"""Development aid: the bitmask NMS path (capacity > 16384) against the oracle on a few shapes, with timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vortex_amd import ops, _lib as L
from oracle import vortex_oracle as O

DEV = torch.device("cuda:0")


def make(B, K, C, seed, n=None, wh=(0.02, 0.2), dup=0, untame=False, ties=0):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, K, 2, generator=g)
    s = torch.rand(B, K, 2, generator=g) * (wh[1] - wh[0]) + wh[0]
    boxes = torch.cat((c - s / 2, c + s / 2), 2).contiguous()
    if dup:
        boxes[:, dup:] = boxes[:, :1]
    if untame:
        boxes[:, 5, 2] = boxes[:, 5, 0] - 0.1
    scores = torch.rand(B, K, generator=g)
    if ties:
        scores = (scores * ties).floor() / ties          # only `ties` distinct scores
    cls = torch.randint(0, C, (B, K), generator=g, dtype=torch.int64)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2).contiguous()
    counts = None
    if n is not None:
        counts = torch.tensor(n, dtype=torch.int32)
    return boxes, scores, cls, det, counts


def check(name, B, K, C, thr=0.45, top_k=None, optimistic=False, **kw):
    boxes, scores, cls, det, counts = make(B, K, C, **kw)
    d = [t.to(DEV) if t is not None else None for t in (boxes, scores, cls, det, counts)]
    keep_idx, out_det, keep_counts, status = ops.batched_nms(d[0], d[1], d[2], d[3], d[4], thr, C, top_k, optimistic=optimistic)
    torch.cuda.synchronize()
    redo = ""
    if optimistic and (int(status.item()) & L.VX_STATUS_DEFERRED):      # the optimistic bet was lost: the full call
        keep_idx, out_det, keep_counts, status = ops.batched_nms(d[0], d[1], d[2], d[3], d[4], thr, C, top_k)
        torch.cuda.synchronize()
        redo = " (deferred -> redone)"
    kc = keep_counts.cpu().tolist()
    ok = True
    for b in range(B):
        n = int(counts[b]) if counts is not None else K
        keep = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k:
            keep = keep[:top_k]
        if kc[b] != keep.numel() or not torch.equal(keep_idx[b, :kc[b]].cpu(), keep):
            ok = False
            got = keep_idx[b, :kc[b]].cpu()
            m = min(got.numel(), keep.numel())
            first = int((got[:m] != keep[:m]).nonzero()[0]) if m and (got[:m] != keep[:m]).any() else m
            print("  image %d: kept %d vs oracle %d, first difference at %d" % (b, kc[b], keep.numel(), first))
        elif not torch.equal(out_det[b, :kc[b]].cpu(), det[b][keep]):
            ok = False
            print("  image %d: rows differ" % b)
    t0 = time.perf_counter()
    for _ in range(5):
        ops.batched_nms(d[0], d[1], d[2], d[3], d[4], thr, C, top_k, optimistic=optimistic)
    torch.cuda.synchronize()
    print("%-40s %s status=%d kept=%s  %.3f ms" % (name + redo, "OK" if ok else "FAIL", int(status.item()), kc[:4], (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    return ok


if __name__ == "__main__":
    allok = True
    allok &= check("1 class, 4500 of cap 20000", 3, 20000, 1, seed=1, n=[4500, 300, 9000])
    allok &= check("1 class optimistic", 3, 20000, 1, seed=2, n=[4500, 100, 8000], optimistic=True)
    allok &= check("1 class, small boxes", 2, 20000, 1, seed=3, n=[6000, 7000], wh=(0.01, 0.05))
    allok &= check("1 class, big boxes", 2, 20000, 1, seed=4, n=[3000, 5000], wh=(0.2, 0.6), thr=0.3)
    allok &= check("3 classes image route", 2, 20000, 3, seed=5, n=[8000, 5000])
    allok &= check("20 classes partitioned 50400", 2, 50400, 20, seed=6)
    allok &= check("20 classes partitioned optimistic", 2, 50400, 20, seed=7, optimistic=True)
    allok &= check("80 classes top_k 300", 2, 100000, 80, seed=8, top_k=300, optimistic=True)
    allok &= check("20 classes top_k 5000", 2, 50400, 20, seed=9, top_k=5000)
    allok &= check("duplicates -> edge overflow", 2, 20000, 1, seed=10, n=[3000, 2000], dup=500)
    allok &= check("untame box -> fallback", 2, 20000, 1, seed=11, n=[3000, 2000], untame=True)
    allok &= check("1 class top_k 100", 2, 20000, 1, seed=12, n=[5000, 2000], top_k=100)
    allok &= check("4 classes 60000 (one class > 16384)", 1, 70000, 4, seed=13)
    allok &= check("prefix: big boxes keep < top_k -> redo", 2, 30000, 1, seed=14, wh=(0.2, 0.6), thr=0.3, top_k=100)
    allok &= check("prefix: same, optimistic 20 classes", 2, 30000, 20, seed=15, wh=(0.3, 0.7), thr=0.2, top_k=600, optimistic=True)
    allok &= check("prefix: 3 distinct scores (no cut)", 2, 30000, 4, seed=16, top_k=50, ties=3)
    allok &= check("prefix: 1000 distinct scores", 2, 60000, 80, seed=17, top_k=300, ties=1000)
    allok &= check("prefix: ragged counts", 3, 40000, 10, seed=18, n=[40000, 3000, 9000], top_k=200)
    allok &= check("prefix: top_k 3000", 2, 100000, 80, seed=19, top_k=3000)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
