"""GPU parity tests of the runtime-side rows: device resize_batch against the
oracle's restatement of cv2.resize's u8 INTER_LINEAR scheme (parity unpinned: cv2 is absent), on-device
output_format slicing + adjust_coordinates against the numpy restatement of the reference's host code."""
import numpy as np
import pytest
import torch

import vortex_amd as V
from vortex_amd import ops, runtime
from oracle import vortex_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

FMT6 = {"bounding_box": {"indices": [0, 1, 2, 3], "axis": 1},
        "class_confidence": {"indices": [4], "axis": 1},
        "class_label": {"indices": [5], "axis": 1}}
FMT16 = {"bounding_box": {"indices": {"start": 0, "end": 4}, "axis": 1},
         "class_confidence": {"indices": [4], "axis": 1},
         "class_label": {"indices": [5], "axis": 1},
         "landmarks": {"indices": {"start": 6, "end": 16}, "axis": 1}}


def _same_results(got, exp):
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert list(a.keys()) == list(b.keys())
        for k in b:
            if b[k] is None:
                assert a[k] is None
            else:
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
                assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("src,dst", [((720, 1280), (416, 416)), ((37, 53), (64, 48)), ((100, 100), (50, 50)),
                                     ((30, 40), (90, 120)), ((416, 416), (416, 416)), ((5, 3), (7, 2)),
                                     # the streaming kernel's corners: 3-pixel-wide source (every window clamped), a
                                     # letterbox offset that is no multiple of 4, one source row, strong up-scaling in x
                                     ((64, 3), (32, 8)), ((100, 37), (64, 64)), ((1, 200), (16, 400)), ((9, 7), (40, 2048))])
@pytest.mark.parametrize("letterbox", [False, True])
def test_resize_u8_vs_oracle(src, dst, letterbox):
    rng = np.random.default_rng(src[0] * 31 + dst[1])
    imgs = rng.integers(0, 256, (2, src[0], src[1], 3), dtype=np.uint8)
    y = ops.resize_u8(torch.from_numpy(imgs).to(DEV), dst, letterbox=letterbox).cpu().numpy()
    ref = np.stack([O.resize_bilinear_u8(im, dst[0], dst[1], letterbox) for im in imgs])
    assert y.dtype == np.uint8 and np.array_equal(y, ref)


def test_resize_batch_mixed_sizes_padding_and_quirk():
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in ((720, 1280), (480, 640), (64, 64))]
    got = runtime.resize_batch(images, (4, 416, 416, 3), device=DEV)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (4, 416, 416, 3)
    assert np.array_equal(got.cpu().numpy(), O.resize_batch(images, (4, 416, 416, 3)))
    assert not bool(got[3].any())
    # letterbox extension
    got = runtime.resize_batch(images, (3, 320, 320, 3), resize_kind="letterbox", device=DEV)
    assert np.array_equal(got.cpu().numpy(), O.resize_batch(images, (3, 320, 320, 3), letterbox=True))
    # a channels-first size tuple goes through the reference's [0,3,1,2] permutation (basic_runtime.py:83),
    # which yields c = H for (n,3,H,W): not a 3-channel request, refused here
    with pytest.raises(AssertionError):
        runtime.resize_batch(images, (3, 3, 320, 320), device=DEV)
    # (h,w) handed to cv2 as dsize=(w,h): w rows, h columns when the batch is full ...
    got = runtime.resize_batch(images, (3, 96, 160, 3), device=DEV)
    assert tuple(got.shape) == (3, 160, 96, 3)
    assert np.array_equal(got.cpu().numpy(), O.resize_batch(images, (3, 96, 160, 3)))
    # ... and np.stack's ValueError when pad images of (h,w) would be mixed in
    with pytest.raises(ValueError):
        O.resize_batch(images, (4, 96, 160, 3))
    with pytest.raises(ValueError):
        runtime.resize_batch(images, (4, 96, 160, 3), device=DEV)


@pytest.mark.parametrize("coordinate_fmt", [None, "relative", "absolute"])
@pytest.mark.parametrize("sthr", [0.06, 0.9])
def test_format_detections_yolo(coordinate_fmt, sthr):
    S, C, B = 416, 80, 5
    pred = O.yolo_head_eval(O.synth_yolo_heads(B, S, C, seed=31), S)
    st, it = torch.tensor([sthr]), torch.tensor([0.2])
    ref = O.yolo_postprocess(pred, st, it, S)
    vis = [np.zeros((h, w, 3), np.uint8) for h, w in ((720, 1280), (333, 500), (416, 416), (1080, 1920), (7, 9))]
    exp = O.get_prediction_results(ref, FMT6)
    if coordinate_fmt is not None:
        exp = O.adjust_coordinates(vis, exp, coordinate_fmt)
    out_det, keep_counts, meta = V.YoloV3PostProcess(S).forward_unpacked(pred.to(DEV), st, it)
    got = runtime.format_detections(out_det, keep_counts, FMT6, vis, coordinate_fmt, status=meta[-1])
    _same_results(got, exp)


def test_format_detections_landmarks_and_empty_images():
    priors = O.make_priors(640)
    x = O.synth_prior_head(3, 50400, 5, 2, seed=8)
    x[1, :, 4 + 10 + 1] = -30.0                              # image 1: no face passes the threshold
    st, it = torch.tensor([0.5]), torch.tensor([0.4])
    pp = V.PriorBoxPostProcess(priors, n_landmarks=5).to(DEV)
    res = pp(x.to(DEV), st, it)
    assert res[1].shape == (0, 16)
    vis = [np.zeros((480, 640, 3), np.uint8), np.zeros((10, 10, 3), np.uint8), np.zeros((1080, 1920, 3), np.uint8)]
    exp = O.adjust_coordinates(vis, O.get_prediction_results([r.cpu() for r in res], FMT16), "relative")
    out_det, keep_counts, meta = pp.forward_unpacked(x.to(DEV), st, it)
    got = runtime.format_detections(out_det, keep_counts, runtime.ResultFormat(FMT16), vis, "relative", status=meta[-1])
    _same_results(got, exp)
    assert got[1]["bounding_box"] is None and got[1]["landmarks"] is None
    assert got[0]["landmarks"].shape[1] == 10
    with pytest.raises(RuntimeError):
        runtime.ResultFormat({"bounding_box": {"indices": [0, 1, 2, 3], "axis": 0}})


def test_inference_helper_end_to_end():
    """InferenceHelper.run_inference (helper.py:226-246): mixed-size images -> resize_batch -> predictor ->
    formatted, pixel-scaled results; compared with the oracle's host-side composition of the same steps."""
    S, C, n = 64, 80, 4
    pred = O.yolo_head_eval(O.synth_yolo_heads(n, S, C, seed=7), S)
    seen = {}

    class Model(torch.nn.Module):
        task = "detection"
        output_format = FMT6

        def forward(self, x):
            seen["x"] = x
            return pred.to(x.device)

    p = V.create_predictor(dict(network=Model(), preprocess=V.Normalize(), postprocess=V.YoloV3PostProcess(S))).to(DEV)
    rng = np.random.default_rng(9)
    images = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in ((120, 200), (64, 64), (90, 31))]
    helper = runtime.InferenceHelper(p, (n, S, S, 3))
    got = helper.run_inference(images, coordinate_fmt="relative", score_threshold=0.06, iou_threshold=0.2)
    batch = O.resize_batch(images, (n, S, S, 3))
    assert np.array_equal(seen["x"].cpu().numpy(), O.normalize(torch.from_numpy(batch), [0.485, 0.456, 0.406],
                                                               [0.229, 0.224, 0.225]).numpy())
    ref = O.yolo_postprocess(pred, torch.tensor([0.06]), torch.tensor([0.2]), S)
    exp = O.adjust_coordinates(images, O.get_prediction_results(ref, FMT6), "relative")
    assert len(got) == n
    _same_results(got, exp)


def test_graphed_postprocess_matches_eager_and_follows_its_inputs():
    """HIP-graph replay (vortex_amd/graphs.py) == eager call, for the data present in the static input tensors at
    replay time; results are copies (a later replay does not change them)."""
    from vortex_amd.graphs import GraphedPostProcess
    S, C, B = 416, 80, 3
    st, it = torch.tensor([0.5]), torch.tensor([0.3])
    h1 = [h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=61)]
    h2 = [h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=62)]
    img = O.synth_images(B, S, seed=5).to(DEV)
    post, norm = V.YoloV3FusedPostProcess(S), V.Normalize().to(DEV)
    static = [h.clone() for h in h1]
    g = GraphedPostProcess(post, static, st, it, preprocess=norm, images=img)
    r1, e1 = g(), post(h1, st, it)
    assert len(r1) == B and all(torch.equal(a, b) for a, b in zip(r1, e1))
    assert torch.equal(g.preprocessed, norm(img))
    for s, h in zip(static, h2):
        s.copy_(h)
    r2, e2 = g(), post(h2, st, it)
    assert all(torch.equal(a, b) for a, b in zip(r2, e2))
    assert all(torch.equal(a, b) for a, b in zip(r1, e1))          # first results untouched by the second replay
    # decoded-head (reference API boundary) variant
    pred = O.yolo_head_eval(O.synth_yolo_heads(2, 64, C, seed=7), 64).to(DEV)
    pp = V.YoloV3PostProcess(64)
    g2 = GraphedPostProcess(pp, pred, 0.06, 0.2)
    assert all(torch.equal(a, b) for a, b in zip(g2(), pp(pred, torch.tensor([0.06]), torch.tensor([0.2]))))


def test_graphed_postprocess_two_launches_in_flight_keep_their_own_results():
    """launch(); new inputs; launch(); launch() -- and only then the results: every PendingDetections returns the
    detections of the inputs ITS launch saw, although the third launch replays into the first one's buffers."""
    from vortex_amd.graphs import GraphedPostProcess
    S, C, B = 416, 80, 3
    st, it = torch.tensor([0.5]), torch.tensor([0.3])
    hs = [[h.to(DEV) for h in O.synth_yolo_heads(B, S, C, seed=70 + k)] for k in range(3)]
    post = V.YoloV3FusedPostProcess(S)
    static = [h.clone() for h in hs[0]]
    g = GraphedPostProcess(post, static, st, it, depth=2)
    pend = []
    for k in range(3):
        for s_, h in zip(static, hs[k]):
            s_.copy_(h)
        pend.append(g.launch())
    for k in (0, 1, 2):
        exp = post(hs[k], st, it)
        got = pend[k].result()
        assert len(got) == B and all(torch.equal(a, b) for a, b in zip(got, exp)), "launch %d returned another batch" % k
