"""The whole-chip NMS of large class segments (vx_nms_bm.h: spatial prep -> vx_k_bm_pairs -> edge-list scan) and the
score-prefix path of top_k on large inputs (vx_k_prefix_thr / _compact), through the C ABI against the oracle
(reference postprocess/utils/nms.py:57-98,105-123 with the oracle's greedy core): keep indices, counts and gathered
rows bit for bit.  Both paths only exist for capacities beyond the single-launch kernel (> 16384 candidates/image).
Covers what can go wrong there: both routes (single-launch kernel / class partition), segments on either side of the
256-box limit, the fall-backs (untame boxes, edge-list overflow by duplicated boxes, a class beyond 16384 boxes),
top_k stop inside the scan, a prefix that keeps too few boxes (redo), score ties at the prefix cut, ragged counts,
the optimistic call with its deferral."""
import pytest
import torch

from oracle import vortex_oracle as O
from vortex_amd import _lib as L, ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def make(B, K, C, seed, n=None, wh=(0.02, 0.2), dup=0, untame=False, ties=0, pixel=False):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, K, 2, generator=g)
    s = torch.rand(B, K, 2, generator=g) * (wh[1] - wh[0]) + wh[0]
    boxes = torch.cat((c - s / 2, c + s / 2), 2).contiguous()
    if pixel:
        boxes = (boxes.clamp(0, 1) * 640).contiguous()
    if dup:
        boxes[:, dup:] = boxes[:, :1]                     # every box behind `dup` is a copy of box 0
    if untame:
        boxes[:, 5, 2] = boxes[:, 5, 0] - 0.1             # inverted
        boxes[:, 9, 3] = boxes[:, 9, 1]                   # zero area
    scores = torch.rand(B, K, generator=g)
    if ties:
        scores = (scores * ties).floor() / ties           # only `ties` distinct scores
    cls = torch.randint(0, C, (B, K), generator=g, dtype=torch.int64)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2).contiguous()
    counts = torch.tensor(n, dtype=torch.int32) if n is not None else None
    return boxes, scores, cls, det, counts


def run(B, K, C, thr=0.45, top_k=None, optimistic=False, expect_deferred=None, **kw):
    boxes, scores, cls, det, counts = make(B, K, C, **kw)
    d = [t.to(DEV) if t is not None else None for t in (boxes, scores, cls, det, counts)]
    keep_idx, out_det, keep_counts, status = ops.batched_nms(*d, thr, C, top_k, optimistic=optimistic)
    st = int(status.item())
    if expect_deferred is not None:
        assert bool(st & L.VX_STATUS_DEFERRED) == expect_deferred
    if st & L.VX_STATUS_DEFERRED:                         # the optimistic bet was lost: the full call (what result() does)
        assert optimistic
        keep_idx, out_det, keep_counts, status = ops.batched_nms(*d, thr, C, top_k)
        st = int(status.item())
    assert ops.status_ok(st), st
    kc = keep_counts.cpu().tolist()
    for b in range(B):
        n = int(counts[b]) if counts is not None else K
        keep = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k:
            keep = keep[:top_k]
        assert kc[b] == keep.numel(), "image %d: kept %d, oracle %d" % (b, kc[b], keep.numel())
        assert torch.equal(keep_idx[b, :kc[b]].cpu(), keep), "keep indices differ in image %d" % b
        assert torch.equal(out_det[b, :kc[b]].cpu(), det[b][keep]), "gathered rows differ in image %d" % b
    return kc


@pytest.mark.parametrize("case", [
    dict(B=3, K=20000, C=1, seed=1, n=[4500, 300, 9000]),                    # image route: one large segment each
    dict(B=3, K=20000, C=1, seed=2, n=[4500, 100, 8000], optimistic=True, expect_deferred=False),
    dict(B=2, K=20000, C=1, seed=3, n=[6000, 7000], wh=(0.01, 0.05)),        # sparse overlaps
    dict(B=2, K=20000, C=1, seed=4, n=[3000, 5000], wh=(0.2, 0.6), thr=0.3), # dense overlaps: long edge lists
    dict(B=2, K=20000, C=3, seed=5, n=[8000, 5000]),                          # several exported segments per image
    dict(B=2, K=20000, C=40, seed=21, n=[16000, 12000]),                      # segments on both sides of 256 boxes
    dict(B=2, K=20000, C=1, seed=22, n=[257, 256]),                           # at the limit
    dict(B=2, K=20000, C=2, seed=23, n=[9000, 700], pixel=True),              # pixel boxes (2049 * cls offsets)
    dict(B=2, K=20000, C=1, seed=24, n=[5000, 4000], thr=0.0),
    dict(B=2, K=20000, C=1, seed=25, n=[5000, 4000], thr=1.0),
    dict(B=2, K=20000, C=1, seed=26, n=[3000, 2000], thr=1e-8),               # threshold outside the bracket's range: sweeps
])
def test_bitmask_image_route(case):
    run(**case)


@pytest.mark.parametrize("case", [
    dict(B=2, K=50400, C=20, seed=6),                                         # class partition: 20 x ~2500 boxes
    dict(B=2, K=50400, C=20, seed=7, optimistic=True, expect_deferred=False),
    dict(B=3, K=40000, C=10, seed=18, n=[40000, 3000, 17000]),                # ragged: both routes in one call
    dict(B=1, K=70000, C=4, seed=13),                                         # one class beyond 16384: general path
    dict(B=2, K=30000, C=200, seed=27),                                       # many small segments + a few large
])
def test_bitmask_partitioned_route(case):
    run(**case)


@pytest.mark.parametrize("case", [
    dict(B=2, K=20000, C=1, seed=10, n=[3000, 2000], dup=500),                # 2500 identical boxes: edge list overflows
    dict(B=2, K=20000, C=1, seed=11, n=[3000, 2000], untame=True),            # inverted / empty boxes: chain sweep
    dict(B=2, K=40000, C=8, seed=28, dup=3000),                               # the same on the partitioned route
    dict(B=2, K=40000, C=8, seed=29, untame=True),
])
def test_bitmask_fallbacks(case):
    run(**case)


@pytest.mark.parametrize("case", [
    dict(B=2, K=100000, C=80, seed=8, top_k=300, optimistic=True, expect_deferred=False),     # BASELINE configs[4] shape
    dict(B=2, K=100000, C=80, seed=19, top_k=3000),
    dict(B=2, K=50400, C=20, seed=9, top_k=5000),                             # prefix target beyond half the list: 8192
    dict(B=2, K=20000, C=1, seed=12, n=[5000, 2000], top_k=100),              # small images: no prefix, top_k stop in the scan
    dict(B=2, K=30000, C=1, seed=14, wh=(0.2, 0.6), thr=0.3, top_k=100),      # the prefix keeps < top_k: redone in the call
    dict(B=2, K=30000, C=20, seed=15, wh=(0.3, 0.7), thr=0.2, top_k=600, optimistic=True, expect_deferred=True),
    dict(B=2, K=30000, C=4, seed=16, top_k=50, ties=3),                       # 3 distinct scores: the cut cannot split a tie
    dict(B=2, K=60000, C=80, seed=17, top_k=300, ties=1000),
    dict(B=3, K=40000, C=10, seed=30, n=[40000, 3000, 9000], top_k=200),
    dict(B=1, K=40000, C=1, seed=31, top_k=1, wh=(0.05, 0.1)),
])
def test_topk_score_prefix(case):
    run(**case)


def test_prefix_equals_full_run_prefix_property():
    """size-independent check at the stress shape: the top_k result is the head of the run without top_k."""
    boxes, scores, cls, det, _ = [t.to(DEV) if t is not None else None for t in make(2, 100000, 80, seed=40)]
    full = ops.batched_nms(boxes, scores, cls, det, None, 0.45, 80, None)
    top = ops.batched_nms(boxes, scores, cls, det, None, 0.45, 80, 300)
    for b in range(2):
        assert int(top[2][b]) == 300 and int(full[2][b]) >= 300
        assert torch.equal(top[0][b, :300], full[0][b, :300])
        assert torch.equal(top[1][b, :300], full[1][b, :300])
