"""The oracle (CPU restatement) against golden vectors produced by the reference's own
modules (tests/golden/gen_golden.py) and against the reference's known-answer tests."""
import numpy as np
import pytest
import torch

from oracle import vortex_oracle as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("cfg", ["imagenet", "half", "unit"])
def test_normalize_bit_exact(golden, cfg):
    g = golden("normalize")
    mean, std = g["mean_" + cfg].tolist(), g["std_" + cfg].tolist()
    for which in ("img", "sweep"):
        x = T(g[which])
        assert np.array_equal(O.normalize(x, mean, std).numpy(), g[f"norm_{which}_{cfg}"])
        assert np.array_equal(O.normalize(x, mean, std, flip=True).numpy(), g[f"flip_{which}_{cfg}"])


@pytest.mark.parametrize("name,grid", [("g13_c80", 13), ("g13_c1", 13), ("g26_c3", 26), ("g52_c2", 52)])
def test_yolo_layer_eval(golden, name, grid):
    g = golden("yolo_layer")
    y = O.yolo_layer_eval(T(g["in_" + name]), 416, grid, g["mask_" + name].tolist())
    # same torch build, same ops, contiguous input -> bit-identical (SURVEY.md §8a3 probe)
    assert np.array_equal(y.numpy(), g["out_" + name])


def test_yolo_layer_ordering_pin(golden):
    """tests/models/test_yolo_head.py:120-156 of the reference: eval centres of row r lie in
    grid cell (x = r % 13, y = (r // 13) % 13) * 32 — anchor-major, then y, then x."""
    g = golden("yolo_layer")
    y = O.yolo_layer_eval(T(g["in_g13_c80"]), 416, 13, [6, 7, 8])
    assert y.shape == (1, 3 * 13 * 13, 85)
    r = torch.arange(507)
    gx, gy = (r % 13).float(), ((r // 13) % 13).float()
    assert torch.all(y[0, :, 0] >= gx * 32) and torch.all(y[0, :, 0] <= (gx + 1) * 32)
    assert torch.all(y[0, :, 1] >= gy * 32) and torch.all(y[0, :, 1] <= (gy + 1) * 32)
    assert torch.all(y[0, :, 4:] >= 0) and torch.all(y[0, :, 4:] <= 1)


@pytest.mark.parametrize("tname,thr", [("t50", 0.5), ("t06", 0.06), ("t00", 0.0), ("t90", 0.9)])
def test_yolo_decoder_bit_exact(golden, tname, thr):
    g = golden("yolo_postprocess")
    pred, S = T(g["pred"]), int(g["img_size"])
    for b in range(pred.shape[0]):
        bb, sc, ci, det = O.yolo_decoder(pred[b:b + 1], torch.tensor([thr]), S)
        k = f"dec_{tname}_b{b}_"
        assert np.array_equal(bb.numpy(), g[k + "bboxes"])
        assert np.array_equal(sc.numpy(), g[k + "scores"])
        assert np.array_equal(ci.numpy(), g[k + "cls"]) and ci.dtype == torch.int64
        assert np.array_equal(det.numpy(), g[k + "det"])


def test_yolo_decoder_no_threshold_and_errors(golden):
    g = golden("yolo_postprocess")
    pred, S = T(g["pred"]), int(g["img_size"])
    bb, sc, ci, det = O.yolo_decoder(pred[0:1], torch.tensor([0.5]), S, threshold=False)
    assert np.array_equal(det.numpy(), g["dec_nothr_b0_det"])
    assert np.array_equal(sc.numpy(), g["dec_nothr_b0_scores"])
    bad = pred[0:1].clone()
    bad[0, 3, 7] = float("nan")
    with pytest.raises(ValueError):
        O.yolo_decoder(bad, torch.tensor([0.5]), S)
    with pytest.raises(RuntimeError):
        O.yolo_decoder(pred[0], torch.tensor([0.5]), S)
    # empty result shapes (SURVEY.md §8a5 probe)
    bb, sc, ci, det = O.yolo_decoder(pred[0:1], torch.tensor([2.0]), S)
    assert bb.shape == (0, 4) and sc.shape == (0,) and ci.shape == (0,) and det.shape == (1, 0, 6)


@pytest.mark.parametrize("pname", ["p06_20", "p00_100", "p90_20", "p001_45", "p00_00"])
def test_postprocess_matches_reference_wrapper(golden, pname):
    g = golden("yolo_postprocess")
    pred, S = T(g["pred"]), int(g["img_size"])
    sthr, ithr = g[f"post_{pname}_thr"]
    res = O.yolo_postprocess(pred, torch.tensor([sthr]), torch.tensor([ithr]), S)
    assert isinstance(res, tuple) and len(res) == pred.shape[0]
    for b, r in enumerate(res):
        assert np.array_equal(r.numpy(), g[f"post_{pname}_b{b}"])


@pytest.mark.parametrize("name", ["k300_c80", "k500_c3", "k64_c1", "k1_c1"])
def test_batched_nms_wrapper(golden, name):
    g = golden("batched_nms")
    boxes, scores, cls = T(g["boxes_" + name]), T(g["scores_" + name]), T(g["cls_" + name])
    thr = torch.tensor(g["thr_" + name])
    keep = O.batched_nms(boxes, scores, cls, thr)
    assert np.array_equal(keep.numpy(), g["keep_" + name])
    res = O.batched_nms_module(T(g["det_" + name]), cls, boxes, scores, thr)
    assert np.array_equal(res.numpy(), g["res_" + name])
    # keep is in descending score order
    s = scores[keep]
    assert torch.all(s[:-1] >= s[1:])


def test_batched_nms_empty(golden):
    g = golden("batched_nms")
    k = O.batched_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64), torch.tensor([0.5]))
    assert k.shape == (0,) and k.dtype == torch.int64 and g["keep_empty"].shape == (0,)


def test_iou_known_answers():
    """Reference known answers: tests/utils/test_prediction.py:29-56."""
    lib = O._lib()

    def f(fn, a, b):
        a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
        return fn(a.ctypes.data, b.ctypes.data)
    a, b, c, d = (1, 1, 4, 4), (5, 1, 8, 4), (1, 5, 4, 8), (2, 6, 5, 9)
    e, e2, f_ = (7, 5, 14, 11), (7, 5, 14, 12), (11, 5, 14, 8)
    I, U = lib.vx_oracle_intersection, lib.vx_oracle_union
    assert f(I, a, a) == 9 and f(I, a, b) == 0 and f(I, c, d) == 4 and f(I, d, c) == 4
    assert f(I, e, f_) == 9 and f(I, f_, e) == 9
    assert f(U, a, a) == 9 and f(U, a, b) == 18 and f(U, c, d) == 14 and f(U, d, c) == 14
    assert f(U, e2, f_) == 49 and f(U, f_, e2) == 49


def test_offset_quantisation_changes_keep_set():
    """the 2049*cls fp32 offset quantises coordinates; exact per-class NMS
    gives a different keep set, so the oracle must (and does) reproduce the quirk."""
    diff = 0
    for seed in range(5):
        boxes, scores, cls = O.synth_stress_boxes(2000, 80, seed)
        boxes = boxes * 0.25 + 0.3          # dense overlaps
        k1 = O.batched_nms(boxes, scores, cls, torch.tensor([0.45]))
        exact = []
        for c in cls.unique():
            idx = torch.nonzero(cls == c).squeeze(1)
            exact.append(idx[O.nms(boxes[idx], scores[idx], 0.45)])
        exact = torch.cat(exact)
        diff += set(k1.tolist()) != set(exact.tolist())
    assert diff > 0


def test_get_prediction_results():
    fmt = {"bounding_box": {"indices": [0, 1, 2, 3], "axis": 1},
           "class_confidence": {"indices": [4], "axis": 1},
           "class_label": {"indices": [5], "axis": 1}}
    r = O.get_prediction_results([torch.arange(12.).view(2, 6), torch.zeros(0, 6)], fmt)
    assert r[0]["bounding_box"].shape == (2, 4) and r[0]["class_label"][1, 0] == 11
    assert r[1]["bounding_box"] is None


def test_resize_identity_and_letterbox():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_bilinear_u8(img, 12, 20), img)
    lb = O.resize_bilinear_u8(img, 16, 16, letterbox=True)
    assert lb.shape == (16, 16, 3) and np.all(lb[:3] == 0) and np.all(lb[-3:] == 0)
    const = np.full((7, 9, 3), 200, np.uint8)
    assert np.all(O.resize_bilinear_u8(const, 16, 16) == 200)
    batch = O.resize_batch([img, const], (4, 16, 16, 3))
    assert batch.shape == (4, 16, 16, 3) and np.all(batch[2:] == 0)


# --------------------------------------------------------------------------- §8f-1 DETR
def test_detr_oracle_equals_reference_golden(golden):
    g = golden("detr")
    x = torch.from_numpy(g["ref_in"])
    res = O.detr_postprocess(x, torch.tensor(0.3))
    assert isinstance(res, tuple) and len(res) == x.shape[0]
    for b, r in enumerate(res):
        assert r.shape[1] == 6
        assert np.array_equal(r.numpy(), g[f"ref_t30_b{b}"])
    for name in ("coco", "small", "wide"):
        x = torch.from_numpy(g["in_" + name])
        for tname, thr in (("t30", 0.3), ("t70", 0.7), ("t00", 0.0), ("t100", 1.0)):
            res = O.detr_postprocess(x, torch.tensor([thr]))
            for b, r in enumerate(res):
                assert np.array_equal(r.numpy(), g[f"{name}_{tname}_b{b}"]), (name, tname, b)
    assert g["coco_t100_b0"].shape == (0, 6)


def test_detr_reference_test_expectation(golden):
    """The expectation of the reference's own test (tests/models/test_detr.py:136-145), recomputed from
    its input tensors: xyxy box | max softmax prob of the foreground classes | label, conf > 0.3."""
    x = torch.from_numpy(golden("detr")["ref_in"])
    det = O.detr_postprocess(x, torch.tensor(0.3))
    for b in range(x.shape[0]):
        bbox, logits = x[b, :, :4], x[b, :, 4:]
        xyxy = torch.stack((bbox[:, 0] - 0.5 * bbox[:, 2], bbox[:, 1] - 0.5 * bbox[:, 3],
                            bbox[:, 0] + 0.5 * bbox[:, 2], bbox[:, 1] + 0.5 * bbox[:, 3]), -1)
        conf, labels = logits.softmax(-1)[:, :-1].max(-1)
        exp = torch.cat((xyxy, conf.unsqueeze(1), labels.float().unsqueeze(1)), -1)[conf > 0.3]
        assert torch.equal(exp, det[b])
    with pytest.raises(RuntimeError):
        O.detr_postprocess(x[0], torch.tensor(0.3))


# --------------------------------------------------------------------------- classification post-process
def test_softmax_oracle_equals_reference_golden(golden):
    g = golden("softmax")
    for name in ("c10", "c1000", "c1", "c65", "ties"):
        out = O.softmax_postprocess(torch.from_numpy(g["in_" + name]))
        assert torch.equal(out, torch.from_numpy(g["out_" + name])), name
    t = g["out_ties"]
    assert t[0, 0] == 5 and t[1, 0] == 0 and t[2, 0] == 0 and t[3, 0] == 129      # first index wins
