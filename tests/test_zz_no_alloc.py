"""vx_batched_nms_noalloc (the allocation-free entry point: static device scratch, no hipMalloc) against
vx_batched_nms through the C ABI AND against the oracle: bit-identical keep indices, counts, gathered rows and status
on two calls in a row, the second one with a smaller input (the shared scratch keeps the first call's contents).
Collected last (test_zz_*): it must pass alone in a fresh process and after every other test."""
import pytest
import torch

from oracle import vortex_oracle as O
from vortex_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _inputs(B, K, C, seed, ragged=True):
    g = torch.Generator(device=DEV).manual_seed(seed)
    c = torch.rand(B, K, 2, generator=g, device=DEV)
    wh = torch.rand(B, K, 2, generator=g, device=DEV) * 0.2 + 0.02
    boxes = torch.cat((c - wh / 2, c + wh / 2), 2).contiguous()
    scores = torch.rand(B, K, generator=g, device=DEV)
    m = min(scores[:, ::7].shape[1], scores[:, 1::7].shape[1])
    scores[:, ::7][:, :m] = scores[:, 1::7][:, :m]                           # some exact score ties
    cls = torch.randint(0, C, (B, K), generator=g, device=DEV, dtype=torch.int64)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2).contiguous()
    counts = torch.randint(0, K + 1, (B,), generator=g, device=DEV, dtype=torch.int32) if ragged else None
    if counts is not None:
        counts[0] = K
        if B > 1:
            counts[1] = 0
    return boxes, scores, cls, det, counts


def _call(noalloc, boxes, scores, cls, det, counts, C, thr, top_k):
    lib = L.lib()
    B, K = scores.shape
    D = det.shape[2]
    out_cap = min(K, top_k) if top_k else K
    keep_idx = torch.full((B, out_cap), -1, dtype=torch.int64, device=DEV)
    out_det = torch.full((B, out_cap, D), -1.0, device=DEV)
    keep_counts = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = L.stream_ptr(DEV)
    cp = counts.data_ptr() if counts is not None else None
    if noalloc:
        rc = lib.vx_batched_nms_noalloc(boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), det.data_ptr(), cp, B, K, D,
                                        C, thr, top_k, keep_idx.data_ptr(), out_det.data_ptr(), keep_counts.data_ptr(),
                                        out_cap, status.data_ptr(), st)
    else:
        nb = lib.vx_batched_nms_workspace_bytes(B, K)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        rc = lib.vx_batched_nms(boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), det.data_ptr(), cp, B, K, D, C,
                                thr, top_k, keep_idx.data_ptr(), out_det.data_ptr(), keep_counts.data_ptr(), out_cap,
                                status.data_ptr(), ws.data_ptr(), nb, st)
    assert rc == 0, L.lib().vx_error_string(rc)
    torch.cuda.synchronize()
    return keep_idx.cpu(), out_det.cpu(), keep_counts.cpu(), int(status.item())


def _same(a, b):
    ka, da, ca, sa = a
    kb, db, cb, sb = b
    assert sa == sb
    assert torch.equal(ca, cb)
    for i in range(ca.numel()):
        n = int(ca[i])
        assert torch.equal(ka[i, :n], kb[i, :n]), "keep indices differ in image %d" % i
        assert torch.equal(da[i, :n], db[i, :n]), "gathered rows differ in image %d" % i


def _same_as_oracle(got, inp, thr, top_k):
    """keep indices and gathered rows of every image against oracle.batched_nms (reference nms.py:57-98,105-123 with the
    oracle's greedy core) on the image's first counts[b] candidates; top_k = the first k of the keep list."""
    boxes, scores, cls, det, counts = [t.cpu() if t is not None else None for t in inp]
    keep_idx, out_det, keep_counts, status = got
    assert status & ~(L.VX_STATUS_RANGE | L.VX_STATUS_RANGE2) == 0
    for b in range(scores.shape[0]):
        n = int(counts[b]) if counts is not None else scores.shape[1]
        keep = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([thr]))
        if top_k:
            keep = keep[:top_k]
        k = int(keep_counts[b])
        assert k == keep.numel(), "image %d: %d kept, oracle %d" % (b, k, keep.numel())
        assert torch.equal(keep_idx[b, :k], keep), "keep indices differ from the oracle in image %d" % b
        assert torch.equal(out_det[b, :k], det[b][keep]), "gathered rows differ from the oracle in image %d" % b


@pytest.mark.parametrize("shapes", [
    # (B, K, classes, top_k): first call, then a smaller one
    [(8, 3000, 20, 0), (3, 700, 20, 0)],                 # single-launch kernel
    [(4, 40000, 80, 300), (2, 20000, 80, 0)],            # class-partitioned multi-kernel path (+ top-k select)
    [(2, 30000, 1, 0), (1, 50, 1, 0)],                   # one class: general path (global sort), then one wave
])
def test_noalloc_matches_batched_nms_two_calls(shapes):
    for n, (B, K, C, top_k) in enumerate(shapes):
        inp = _inputs(B, K, C, seed=100 + n)
        ref = _call(False, *inp, C, 0.45, top_k)
        got = _call(True, *inp, C, 0.45, top_k)
        assert int(ref[2].sum()) > 0
        _same(got, ref)
        _same_as_oracle(got, inp, 0.45, top_k)


def test_noalloc_rejects_oversized_calls_and_allocates_nothing():
    lib = L.lib()
    assert lib.vx_batched_nms_noalloc_max_bytes() == 64 << 20
    p = 1 << 20
    # 256 images x 16384 padded candidates need ~150 MB of scratch: refused on the host, nothing is launched
    assert lib.vx_batched_nms_workspace_bytes(256, 10647) > lib.vx_batched_nms_noalloc_max_bytes()
    assert lib.vx_batched_nms_noalloc(p, p, p, p, None, 256, 10647, 6, 80, 0.5, 0, p, p, p, 10647, p, None) == -6
    assert b"static scratch" in lib.vx_error_string(-6)
    # no device allocation inside the call (torch's allocator is not involved; the HIP pool must not grow either)
    inp = _inputs(4, 2000, 20, seed=5)
    _call(True, *inp, 20, 0.5, 0)                      # set-up call (symbol address lookup)
    B, K = inp[1].shape
    keep_idx = torch.empty((B, K), dtype=torch.int64, device=DEV)
    out_det = torch.empty((B, K, 6), device=DEV)
    kc = torch.empty(B, dtype=torch.int32, device=DEV)
    stt = torch.zeros(1, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(3):
        rc = lib.vx_batched_nms_noalloc(inp[0].data_ptr(), inp[1].data_ptr(), inp[2].data_ptr(), inp[3].data_ptr(),
                                        inp[4].data_ptr(), B, K, 6, 20, 0.5, 0, keep_idx.data_ptr(), out_det.data_ptr(),
                                        kc.data_ptr(), K, stt.data_ptr(), L.stream_ptr(DEV))
        assert rc == 0
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    assert free2 >= free1 - (1 << 20), "device memory shrank across vx_batched_nms_noalloc calls"
