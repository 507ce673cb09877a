"""StreamRing: steps enqueued on two HIP streams in turn, several in flight, give the bits of the synchronous calls
(distinct streams + distinct workspaces/outputs per call is all the library asks for: INTEGRATION.md)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nstreams", [2, 3])
def test_prior_postprocess_on_several_streams_equals_synchronous_calls(nstreams):
    import vortex_amd as V
    from tools.bench_configs import make_priors
    dev = torch.device("cuda", 0)
    priors = make_priors(640).to(dev)
    P, Lm, B = priors.size(0), 5, 6
    pp = V.PriorBoxPostProcess(priors, n_landmarks=Lm).to(dev)
    norm = V.Normalize([0.5] * 3, [0.5] * 3).to(dev)
    st, it = torch.tensor([0.02]), torch.tensor([0.4])
    g = torch.Generator(device=dev).manual_seed(77)
    xs, imgs = [], []
    for _ in range(6):
        x = torch.randn(B, P, 16, generator=g, device=dev)
        x[..., 15:] = x[..., 15:] * 2 - 3
        xs.append(x)
        imgs.append(torch.randint(0, 256, (B, 96, 96, 3), generator=g, dtype=torch.uint8, device=dev))
    want = [(norm(i), pp(x, st, it)) for i, x in zip(imgs, xs)]
    ring = V.StreamRing(nstreams, dev)
    assert len(ring) == nstreams and len(set(s_.cuda_stream for s_ in ring.streams)) == nstreams
    pend = []
    for i, x in zip(imgs, xs):                              # all six in flight before the first result is read
        with ring.next():
            pend.append((norm(i), pp.forward_async(x, st, it)))
    for (n2, p2), (n1, r1) in zip(pend, want):
        r2 = p2.result()
        torch.cuda.synchronize()
        assert torch.equal(n1, n2)
        assert len(r1) == len(r2) == B
        for a, b in zip(r1, r2):
            assert torch.equal(a, b)


def test_yolo_fused_postprocess_on_two_streams_equals_synchronous_calls():
    import bench
    import vortex_amd as V
    dev = torch.device("cuda", 0)
    post = V.YoloV3FusedPostProcess(img_size=416)
    st, it = torch.tensor([0.3]), torch.tensor([0.45])
    heads = [bench.synth_on_device(4, dev, seed=40 + k)[1] for k in range(5)]
    want = [post(h, st, it) for h in heads]
    ring = V.StreamRing(2, dev)
    pend = []
    for h in heads:
        with ring.next():
            pend.append(post.forward_async(h, st, it))
    for p, w in zip(pend, want):
        r = p.result()
        assert len(r) == len(w)
        for a, b in zip(r, w):
            assert torch.equal(a, b)


def test_ring_members_overlap_pairwise():
    """The ring's streams were chosen with vx_streams_overlap: asking again gives the same answer, and a stream paired
    with itself does not overlap (the probe can tell).  The probe times kernels, so each question is asked up to three
    times; a GPU on which no overlapping pair was found (ring.overlapping False) skips the first half."""
    import ctypes
    import vortex_amd as V
    from vortex_amd import _lib as L
    dev = torch.device("cuda", 0)
    ring = V.StreamRing(2, dev)
    assert V.StreamRing(2, dev).streams == ring.streams                  # chosen once per device
    a, b = ring.streams
    assert a.cuda_stream != b.cuda_stream

    def probe(x, y):
        ov, t1, t2 = ctypes.c_int(-1), ctypes.c_float(0), ctypes.c_float(0)
        L.check(L.lib().vx_streams_overlap(x.cuda_stream, y.cuda_stream, ctypes.byref(ov), ctypes.byref(t1), ctypes.byref(t2)), "probe")
        assert ov.value in (0, 1) and t1.value > 0 and t2.value > 0
        assert (ov.value == 1) == (t2.value < 1.5 * t1.value)
        return ov.value
    if ring.overlapping:
        assert any(probe(a, b) == 1 for _ in range(3))
    assert any(probe(a, a) == 0 for _ in range(3))
    assert L.lib().vx_streams_overlap(a.cuda_stream, b.cuda_stream, None, None, None) == -1
