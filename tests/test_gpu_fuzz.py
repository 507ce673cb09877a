"""Seeded fuzz parity: random sizes, class counts, thresholds and DEGENERATE inputs (ties, duplicates, zero-area and
inverted boxes, quantised scores, ragged batches) through the C ABI against the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

import vortex_amd as V
from vortex_amd import ops
from oracle import vortex_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _boxes(kind, K, g):
    if kind == "normal":
        c, wh = torch.rand(K, 2, generator=g), torch.rand(K, 2, generator=g) * 0.3 + 0.01
        return torch.cat((c - wh / 2, c + wh / 2), 1)
    if kind == "grid":                                   # integer grid / 16: exact IoU ties at the threshold
        xy = torch.randint(0, 12, (K, 2), generator=g).float() / 16
        wh = torch.randint(0, 6, (K, 2), generator=g).float() / 16          # width 0 allowed: zero-area boxes
        return torch.cat((xy, xy + wh), 1)
    if kind == "dup":                                    # few distinct boxes repeated many times
        base = _boxes("normal", max(1, K // 20), g)
        return base[torch.randint(0, base.shape[0], (K,), generator=g)]
    if kind == "inverted":                               # x2 < x1 for some boxes (negative width -> negative area)
        b = _boxes("normal", K, g)
        flip = torch.rand(K, generator=g) < 0.2
        b[flip] = b[flip][:, [2, 1, 0, 3]]
        return b
    raise AssertionError(kind)


def _case(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    K = int(torch.randint(1, 3000, (1,), generator=g))
    C = [1, 2, 5, 80, 1000, 4096][seed % 6]
    kind = ["normal", "grid", "dup", "inverted"][seed % 4]
    boxes = _boxes(kind, K, g).contiguous()
    scores = torch.rand(K, generator=g)
    if seed % 3 == 0:
        scores = (scores * 8).floor() / 8                # heavy score ties: order decided by the index
    cls = torch.randint(0, C, (K,), generator=g)
    thr = [0.0, 0.1, 0.25, 0.3333333, 0.45, 0.5, 0.75, 1.0][seed % 8]
    return boxes, scores, cls, thr


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_batched_nms(seed):
    boxes, scores, cls, thr = _case(seed)
    ref = O.batched_nms(boxes, scores, cls, torch.tensor([thr]))
    got = V.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), torch.tensor([thr]))
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), ref), (seed, thr, boxes.shape)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_declared_classes_beyond_the_single_launch_kernel(seed):
    """Capacities beyond 16384 with the class count declared: class-partitioned sort + sweep (chain, spatial or wave
    routine per segment; inverted / zero-area boxes are not "tame" and must fall back to the chain sweep), kept keys as
    ordered runs + merge rounds (no top_k, or a top_k beyond the radix select) or the select (small top_k), two images
    of different fill per call."""
    g = torch.Generator().manual_seed(7000 + seed)
    K = int(torch.randint(17000, 45000, (1,), generator=g))
    C = [4, 20, 80, 256][seed % 4]
    kind = ["normal", "grid", "dup", "inverted"][(seed // 2) % 4]
    top_k = [None, 3000, 100][seed % 3]
    thr = [0.3, 0.45, 0.6][seed % 3]
    B = 2
    counts = torch.tensor([K, int(K * 0.6)], dtype=torch.int32)
    boxes = torch.stack([_boxes(kind, K, g) for _ in range(B)]).contiguous()
    scores = torch.rand(B, K, generator=g)
    if seed % 4 == 1:
        scores = (scores * 64).floor() / 64
    cls = torch.randint(0, C, (B, K), generator=g)
    if seed % 5 == 0:                                    # one dominant class: a segment beyond the partitioned path
        cls[0, : int(K * 0.8)] = C - 1
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV),
                                                             counts.to(DEV), thr, n_classes=C, top_k=top_k)
    assert ops.status_ok(int(status.item()))
    for b in range(B):
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k:
            ref = ref[:top_k]
        kc = int(keep_counts[b].item())
        assert kc == ref.numel(), (seed, b, kc, ref.numel())
        assert torch.equal(keep_idx[b, :kc].cpu(), ref), (seed, b)
        assert torch.equal(out_det[b, :kc].cpu(), det[b, ref]), (seed, b)


@pytest.mark.parametrize("K", [17000, 33000])
def test_highest_class_index_on_the_multi_kernel_path(K):
    """Class 4095 (the last one a key can hold) with more candidates than the single-launch kernel sorts: its
    segment has no successor class to search for (found by tests/fuzz_campaign.py)."""
    g = torch.Generator().manual_seed(K)
    boxes = _boxes("normal", K, g).contiguous()
    scores = torch.rand(K, generator=g)
    cls = torch.randint(4000, 4096, (K,), generator=g)
    cls[::7] = 4095
    ref = O.batched_nms(boxes, scores, cls, torch.tensor([0.45]))
    got = V.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), torch.tensor([0.45]))
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("cap", [40, 700, 17000])
def test_counts_outside_the_capacity_are_clamped(cap):
    """A caller-supplied count below 0 reads as 0, above the capacity as the capacity (no out-of-range access),
    on the single-wave, single-launch and multi-kernel paths, and in the packing / formatting kernels."""
    g = torch.Generator().manual_seed(cap)
    B, C = 3, 5
    boxes = torch.stack([_boxes("normal", cap, g) for _ in range(B)]).contiguous()
    scores = torch.rand(B, cap, generator=g)
    cls = torch.randint(0, C, (B, cap), generator=g)
    det = torch.randn(B, cap, 6, generator=g)
    counts = torch.tensor([-3, cap + 100, min(5, cap)], dtype=torch.int32)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(
        boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), counts.to(DEV), 0.5, C, None)
    assert ops.status_ok(int(status.item()))
    kc = keep_counts.cpu().tolist()
    for b, n in enumerate([0, cap, min(5, cap)]):
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([0.5]))
        assert kc[b] == ref.numel() and torch.equal(keep_idx[b, :kc[b]].cpu(), ref)
    packed = ops.pack_rows(det.to(DEV), counts.to(DEV)).cpu()
    assert torch.equal(packed[:cap + min(5, cap)], torch.cat((det[1], det[2, :min(5, cap)])))
    rows = ops.format_results(det.to(DEV), counts.to(DEV), [0, 5]).cpu()
    assert torch.equal(rows[:cap + min(5, cap)], torch.cat((det[1], det[2, :min(5, cap)]))[:, [0, 5]])


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_ragged_batch_with_top_k(seed):
    """One launch, B images with different candidate counts (incl. 0 and 1) and a top_k; each image against the
    oracle run on its own slice."""
    g = torch.Generator().manual_seed(77 + seed)
    B, cap, C = 9, 700, [3, 80][seed % 2]
    counts = torch.randint(0, cap + 1, (B,), generator=g)
    counts[0], counts[1], counts[2] = 0, 1, cap
    boxes = torch.stack([_boxes(["normal", "grid", "dup"][b % 3], cap, g) for b in range(B)]).contiguous()
    scores = torch.rand(B, cap, generator=g)
    cls = torch.randint(0, C, (B, cap), generator=g)
    det = torch.randn(B, cap, 7, generator=g)
    top_k = [None, 50][seed % 2]
    thr = [0.3, 0.5, 0.0][seed % 3]
    keep_idx, out_det, keep_counts, status = ops.batched_nms(
        boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), counts.to(torch.int32).to(DEV), thr, C, top_k)
    assert ops.status_ok(int(status.item()))
    kc = keep_counts.cpu().tolist()
    for b in range(B):
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k is not None:
            ref = ref[:top_k]
        assert kc[b] == ref.numel(), (b, n)
        assert torch.equal(keep_idx[b, :kc[b]].cpu(), ref)
        assert torch.equal(out_det[b, :kc[b]].cpu(), det[b, :n][ref])


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_yolo_postprocess(seed):
    """Decoded-head boundary (bit-exact): random image size / class count / thresholds, rows with objectness exactly
    at the threshold (strict '>'), all-below-threshold images."""
    g = torch.Generator().manual_seed(300 + seed)
    S = [64, 96, 128, 160][seed % 4]
    C = [1, 3, 20, 80][(seed // 2) % 4]
    B = 1 + seed % 3
    pred = O.yolo_head_eval(O.synth_yolo_heads(B, S, C, seed=500 + seed), S)
    sthr = [0.06, 0.5, 0.9, 0.0][seed % 4]
    ithr = [0.2, 0.45, 1.0, 0.0][(seed // 3) % 4]
    R = pred.shape[1]
    idx = torch.randint(0, R, (R // 10,), generator=g)
    pred[0, idx, 4] = sthr                               # exactly at the threshold: must be dropped
    if B > 1:
        pred[1, :, 4] = 0.0                              # nothing survives in image 1 (unless sthr < 0)
    st, it = torch.tensor([sthr]), torch.tensor([ithr])
    ref = O.yolo_postprocess(pred, st, it, S)
    got = V.YoloV3PostProcess(S)(pred.to(DEV), st, it)
    assert len(got) == B
    for a, e in zip(got, ref):
        assert a.shape == e.shape and torch.equal(a.cpu(), e)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_fused_decode_modes(seed):
    """Raw-head boundary: streaming decode == threshold-first decode == YoloV3Layer x3 + decoder, bit for bit, for
    odd grid sizes (G*G % 4 != 0), tiny and large class counts and thresholds from dense to empty."""
    S = [32, 96, 160, 224, 416, 608, 352, 64, 480, 288][seed]
    C = [1, 2, 7, 80, 80, 20, 243, 3, 5, 91][seed]      # 243 = the K2/K3 limit (5+C <= 248)
    B = [3, 2, 2, 2, 1, 1, 1, 4, 1, 2][seed]
    thr = [0.5, 0.06, 0.9, 0.3, 0.7, 0.95, 0.2, 0.0, 0.999, 0.5][seed]
    heads = [h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=900 + seed)]
    a = ops.yolo_fused_decode(heads, S, thr)
    t = ops.yolo_fused_decode(heads, S, thr, threshold_first=True)
    out = torch.empty((B, ops.yolo_rows(S), 5 + C), device=DEV)
    off = 0
    for h, mask in zip(heads, O.ANCHOR_MASKS):
        G = h.shape[-1]
        V.YoloV3Layer(S, (G, G), list(mask), O.COCO_ANCHORS, C).to(DEV).eval()(h, out=out, row_offset=off)
        off += 3 * G * G
    u = ops.yolo_decode(out, S, thr)
    assert torch.equal(a.meta, t.meta) and torch.equal(a.meta, u.meta)
    for b, k in enumerate(a.counts.cpu().tolist()):
        for x, y, z in ((a.boxes, t.boxes, u.boxes), (a.scores, t.scores, u.scores), (a.cls, t.cls, u.cls), (a.det, t.det, u.det)):
            assert torch.equal(x[b, :k], y[b, :k]) and torch.equal(x[b, :k], z[b, :k])


def test_optimistic_nms_small_and_deferred_608():
    """S=608: 22743 rows per image exceed the single-launch NMS kernel's 16384.  A sparse batch is finished by one
    launch (vx_batched_nms_small); a batch with one dense image reports VX_STATUS_DEFERRED and is redone by the
    general path -- both equal the layer-by-layer path and the oracle's NMS on the GPU's candidates."""
    S, C, B = 608, 80, 2
    heads = O.synth_yolo_heads(B, S, C, seed=77)
    dheads = [h.to(DEV) for h in heads]
    it = torch.tensor([0.45])
    post = V.YoloV3FusedPostProcess(S)
    # 80 declared classes: the optimistic call also takes the class-partitioned path, so the dense batch (22 743
    # candidates per image, no class beyond 16384) is finished without a redo as well
    for sthr, expect_deferred in ((0.9, False), (0.001, False)):
        st = torch.tensor([sthr])
        pend = post.forward_async(dheads, st, it)
        res = pend.result()
        deferred = bool(pend._host[-1].item() & V._lib.VX_STATUS_DEFERRED)
        assert deferred == expect_deferred
        c = ops.yolo_fused_decode(dheads, S, sthr)
        for b, k in enumerate(c.counts.cpu().tolist()):
            keep = O.batched_nms(c.boxes[b, :k].cpu(), c.scores[b, :k].cpu(), c.cls[b, :k].cpu(), it)
            assert torch.equal(res[b].cpu(), c.det[b, :k].cpu()[keep])


@pytest.mark.parametrize("NL", [21, 2])
def test_prior_postprocess_async_with_and_without_the_optimistic_launch(NL):
    """PriorBoxPostProcess.forward_async at P = 50400 with images beyond 16384 candidates.  20 classes: the optimistic
    call finishes them through the class-partitioned path.  One class (RetinaFace layout): the bet is lost
    (VX_STATUS_DEFERRED) and result() redoes the batch; with `optimistic_nms = False` the general path is enqueued
    directly and nothing is deferred.  Same tuples either way."""
    P, L, B = 50400, 0, 2
    priors = O.make_priors(640)
    x = O.synth_prior_head(B, P, L, NL, seed=91)
    x[..., 5:] = x[..., 5:] * 2 - 1                       # most rows pass: ~30k candidates per image
    if NL == 2:
        x[..., 5] = x[..., 5].abs() + 0.5                 # face logit above the background logit for most rows
        x[..., 4] = -x[..., 4].abs()
    st, it = torch.tensor([0.3]), torch.tensor([0.4])
    pp = V.PriorBoxPostProcess(priors, n_landmarks=L).to(DEV)
    xd = x.to(DEV)
    assert int(ops.prior_decode(xd, priors.to(DEV), L, float(st)).counts.max()) > 16384
    ref = pp(xd, st, it)
    assert max(int(r.shape[0]) for r in ref) > 0
    pend = pp.forward_async(xd, st, it)
    res = pend.result()
    assert bool(pend._host[-1].item() & V._lib.VX_STATUS_DEFERRED) == (NL == 2)
    pp.optimistic_nms = False
    pend2 = pp.forward_async(xd, st, it)
    res2 = pend2.result()
    assert not bool(pend2._host[-1].item() & V._lib.VX_STATUS_DEFERRED)
    for a, b, c in zip(ref, res, res2):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_optimistic_call_defers_an_image_with_an_oversized_class():
    """vx_batched_nms_small with 20 declared classes: image 0 (balanced classes, 30 000 boxes) is finished by the
    class-partitioned path, image 1 has one class of ~19 000 boxes and is reported through VX_STATUS_DEFERRED with a keep
    count of 0; vx_batched_nms then finishes both."""
    K, C = 30000, 20
    g = torch.Generator().manual_seed(5150)
    boxes = torch.stack([_boxes("normal", K, g) for _ in range(2)]).contiguous()
    scores = torch.rand(2, K, generator=g)
    cls = torch.randint(0, C, (2, K), generator=g)
    cls[1, : int(K * 0.64)] = 3
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2)
    args = (boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), None, 0.45)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(*args, n_classes=C, optimistic=True)
    assert int(status.item()) & V._lib.VX_STATUS_DEFERRED
    ref0 = O.batched_nms(boxes[0], scores[0], cls[0], torch.tensor([0.45]))
    kc = keep_counts.cpu().tolist()
    assert kc[0] == ref0.numel() and kc[1] == 0
    assert torch.equal(keep_idx[0, :kc[0]].cpu(), ref0)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(*args, n_classes=C)
    assert ops.status_ok(int(status.item()))
    for b in range(2):
        ref = O.batched_nms(boxes[b], scores[b], cls[b], torch.tensor([0.45]))
        assert int(keep_counts[b]) == ref.numel() and torch.equal(keep_idx[b, :ref.numel()].cpu(), ref)


def test_tail_stream_pipeline_equals_single_stream():
    """vx_yolo_fused_postprocess_overlapped (NMS + packing on a second stream behind the decode) returns the same
    tuples as the single-stream call, batch after batch with results collected one batch late."""
    S, C, B = 416, 80, 4
    tail = torch.cuda.Stream(DEV)
    one, two = V.YoloV3FusedPostProcess(S), V.YoloV3FusedPostProcess(S, tail_stream=tail)
    it = torch.tensor([0.45])
    prev = None
    for step, sthr in enumerate((0.9, 0.3, 0.06, 0.9, 0.5)):
        heads = [h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=40 + step)]
        st = torch.tensor([sthr])
        ref = one(heads, st, it)
        pend = two.forward_async(heads, st, it)
        if prev is not None:
            got, exp = prev[0].result(), prev[1]
            assert len(got) == len(exp) and all(torch.equal(a, b) for a, b in zip(got, exp))
        prev = (pend, ref)
    got, exp = prev[0].result(), prev[1]
    assert all(torch.equal(a, b) for a, b in zip(got, exp))


@pytest.mark.parametrize("C,K,skew", [(5, 40000, True), (5, 40000, False), (256, 30000, False), (4, 17000, True),
                                      (80, 20000, False)])
def test_class_partitioned_path_and_its_fallback(C, K, skew):
    """Declared class counts of 4..256 with more candidates than the single-launch kernel take the class-partitioned
    path (count, scatter, per-class LDS sort + sweep); an image with one class beyond 16384 boxes (skew) falls back
    to the general path inside the same call, next to an image that does not."""
    g = torch.Generator().manual_seed(C * 1000 + K)
    B = 2
    boxes = torch.stack([_boxes("normal", K, g) for _ in range(B)]).contiguous()
    scores = torch.rand(B, K, generator=g)
    cls = torch.randint(0, C, (B, K), generator=g)
    if skew:
        cls[0, : int(K * 0.9)] = C - 1                    # image 0: one segment of 0.9 K boxes; image 1 stays balanced
    det = torch.randn(B, K, 6, generator=g)
    counts = torch.tensor([K, K - 123], dtype=torch.int32)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(
        boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), counts.to(DEV), 0.5, C, None)
    assert ops.status_ok(int(status.item()))
    kc = keep_counts.cpu().tolist()
    for b in range(B):
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([0.5]))
        assert kc[b] == ref.numel() and torch.equal(keep_idx[b, :kc[b]].cpu(), ref)
        assert torch.equal(out_det[b, :kc[b]].cpu(), det[b, :n][ref])


def test_class_index_beyond_the_declared_count_is_reported():
    g = torch.Generator().manual_seed(5)
    for K in (50, 3000, 20000):                           # single-wave, single-launch, partitioned paths
        boxes = _boxes("normal", K, g)[None].contiguous()
        scores = torch.rand(1, K, generator=g)
        cls = torch.randint(0, 10, (1, K), generator=g)
        cls[0, K // 2] = 10                               # declared: 10 classes
        det = torch.randn(1, K, 6, generator=g)
        *_, status = ops.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), None, 0.5, 10, None)
        assert int(status.item()) & V._lib.VX_STATUS_BAD_CLASS


@pytest.mark.parametrize("levels", [4, 1 << 20])
@pytest.mark.parametrize("top_k", [1, 300, 2048, 5000])
def test_top_k_select_with_tied_scores(levels, top_k):
    """top_k on the multi-kernel path: 30 000 disjoint boxes (all kept), scores quantised to `levels` values -- with 4
    levels the radix select has to descend into the index bits of the keys; top_k beyond the select's limit takes the
    full second sort."""
    g = torch.Generator().manual_seed(levels % 97 + top_k)
    K, C = 30000, 3
    gx = (torch.arange(K) % 200).float() * 0.005
    gy = (torch.arange(K) // 200).float() * 0.005
    boxes = torch.stack((gx, gy, gx + 0.004, gy + 0.004), 1)[torch.randperm(K, generator=g)][None].contiguous()
    scores = ((torch.rand(1, K, generator=g) * levels).floor() / levels).contiguous()
    cls = torch.randint(0, C, (1, K), generator=g)
    det = torch.randn(1, K, 6, generator=g)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(
        boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), None, 0.5, C, top_k)
    ref = O.batched_nms(boxes[0], scores[0], cls[0], torch.tensor([0.5]))
    assert ref.numel() == K                                # nothing overlaps
    ref = ref[:top_k]
    assert ops.status_ok(int(status.item())) and int(keep_counts[0]) == ref.numel()
    assert torch.equal(keep_idx[0, :ref.numel()].cpu(), ref)
    assert torch.equal(out_det[0, :ref.numel()].cpu(), det[0][ref])
