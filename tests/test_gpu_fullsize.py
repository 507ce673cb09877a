"""GPU tests at BASELINE.json's full configuration sizes.  The oracle checks a sample of the images
bit for bit; every image is covered by size-independent properties: fused == unfused, descending
scores, idempotence (NMS of the kept set keeps everything), no surviving same-class pair above the
IoU threshold, top-k prefix property, counts conservation."""
import numpy as np
import pytest
import torch

import vortex_amd as V
from vortex_amd import ops
from oracle import vortex_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def synth_heads_gpu(B, S, C, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    heads = []
    for G in (S // 32, S // 16, S // 8):
        t = torch.randn(B, 3, 5 + C, G, G, generator=g, device=DEV)
        t[:, :, 2:4] *= 0.5
        t[:, :, 4] = t[:, :, 4] * 2 - 4
        t[:, :, 5:] = t[:, :, 5:] * 1.5 - 2
        heads.append(t.view(B, 3 * (5 + C), G, G).contiguous())
    return heads


def pair_iou_offset(boxes, cls):
    """IoU matrix on the class-offset boxes, same operation order as the reference / oracle."""
    b = boxes + (cls.to(torch.float32) * 2049.0)[:, None]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    xx1 = torch.maximum(b[:, None, 0], b[None, :, 0]); yy1 = torch.maximum(b[:, None, 1], b[None, :, 1])
    xx2 = torch.minimum(b[:, None, 2], b[None, :, 2]); yy2 = torch.minimum(b[:, None, 3], b[None, :, 3])
    inter = (xx2 - xx1).clamp_min(0) * (yy2 - yy1).clamp_min(0)
    return inter / (area[:, None] + area[None, :] - inter)


def check_kept_properties(det_rows, scores, cls, iou_thr):
    """det_rows f32[K',>=6] kept rows of one image, scores/cls of those rows (score order)."""
    if det_rows.shape[0] < 2:
        return
    assert bool((scores[:-1] >= scores[1:]).all()), "keep list not in descending score order"
    iou = pair_iou_offset(det_rows[:, :4], cls)
    iou = torch.triu(iou, diagonal=1)
    assert not bool((iou > iou_thr).any()), "two kept boxes of one class overlap above the threshold"


@pytest.mark.parametrize("sthr,ithr", [(0.9, 0.2), (0.06, 0.2)])
def test_yolov3_416_batch256(sthr, ithr):
    """BASELINE configs[3]: YOLOv3 416x416, batch 256, 80 classes, synthetic heads."""
    S, C, B = 416, 80, 256
    heads = synth_heads_gpu(B, S, C, seed=77)
    st, it = torch.tensor([sthr]), torch.tensor([ithr])
    cand = ops.yolo_fused_decode(heads, S, sthr)
    fused = V.YoloV3FusedPostProcess(img_size=S)(heads, st, it)
    assert len(fused) == B
    # unfused path through the decoded [B,10647,85] tensor must give the same bits
    out = torch.empty((B, ops.yolo_rows(S), 5 + C), device=DEV)
    off = 0
    for h, mask in zip(heads, O.ANCHOR_MASKS):
        G = h.shape[-1]
        V.YoloV3Layer(S, (G, G), list(mask), O.COCO_ANCHORS, C).to(DEV).eval()(h, out=out, row_offset=off)
        off += 3 * G * G
    unfused = V.YoloV3PostProcess(img_size=S)(out, st, it)
    counts = cand.counts.cpu().tolist()
    for b in range(B):
        assert torch.equal(fused[b], unfused[b])
        assert fused[b].shape[0] <= counts[b]
    # decoded tensor vs the oracle's head decode on a sample (tolerance class)
    sample = [0, 1, 100, 255]
    ref_pred = O.yolo_head_eval([h[sample].cpu() for h in heads], S)
    got = out[sample].cpu()
    assert torch.allclose(got, ref_pred, rtol=1e-5, atol=1e-6)
    # decoder + NMS from the GPU's decoded tensor: bit-exact against the oracle on the sample
    ref = O.yolo_postprocess(got, st, it, S)
    for i, b in enumerate(sample):
        assert torch.equal(unfused[b].cpu(), ref[i])
    # properties on every image: order, suppression, idempotence
    post = V.BatchedNMS()
    for b in range(0, B, 7):
        r = fused[b]
        if r.shape[0] == 0:
            continue
        k = counts[b]
        # score of a kept row: find it among the candidates via its (unique) box+conf row
        cb, cs, cc, cd = cand.boxes[b, :k], cand.scores[b, :k], cand.cls[b, :k], cand.det[b, :k]
        keep = V.batched_nms(cb, cs, cc, it)
        assert torch.equal(cd[keep], r)
        check_kept_properties(r, cs[keep], cc[keep], ithr)
        again = post(detections=r[None], class_indexes=cc[keep], bboxes=cb[keep], scores=cs[keep], iou_threshold=it)
        assert torch.equal(again[0], r), "NMS is not idempotent on its own output"


@pytest.mark.timeout(400)
def test_nms_stress_100k_boxes_80_classes_topk300():
    """BASELINE configs[4]: 100k candidate boxes / image, 80 classes, score_thr 0.001, top-k 300, batch 32."""
    B, K, C, ithr, topk = 32, 100_000, 80, 0.45, 300
    boxes, scores, cls = [], [], []
    for b in range(B):
        bb, ss, cc = O.synth_stress_boxes(K, C, seed=1000 + b)
        boxes.append(bb); scores.append(ss); cls.append(cc)
    boxes, scores, cls = torch.stack(boxes), torch.stack(scores), torch.stack(cls)
    counts = (scores > 0.001).sum(1)                      # score threshold: compact on the host side of the test
    order = torch.argsort((scores <= 0.001).to(torch.int8), dim=1, stable=True)   # survivors first, original order kept
    boxes = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4)).contiguous()
    scores = torch.gather(scores, 1, order).contiguous()
    cls = torch.gather(cls, 1, order).contiguous()
    det = torch.cat((boxes, scores[:, :, None], cls[:, :, None].float()), 2).contiguous()
    dcounts = counts.to(torch.int32).to(DEV)
    args = (boxes.to(DEV), scores.to(DEV), cls.to(DEV), det.to(DEV), dcounts)
    keep_idx, out_det, keep_counts, status = ops.batched_nms(*args, ithr, C, topk)
    keep_all, _, keep_counts_all, _ = ops.batched_nms(*args, ithr, C, None)
    kc, kca = keep_counts.cpu().tolist(), keep_counts_all.cpu().tolist()
    assert ops.status_ok(int(status.item()))
    for b in range(B):
        assert kc[b] == min(topk, kca[b])
        assert torch.equal(keep_idx[b, :kc[b]], keep_all[b, :kc[b]]), "top-k is not a prefix of the full keep list"
        ki = keep_idx[b, :kc[b]].cpu()
        assert torch.equal(out_det[b, :kc[b]].cpu(), det[b][ki])
        check_kept_properties(det[b][ki], scores[b][ki], cls[b][ki], ithr)
    for b in (31,):                                       # bit-exact keep list against the oracle (~35 s of CPU)
        n = int(counts[b])
        ref = O.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], torch.tensor([ithr]))
        assert torch.equal(keep_all[b, :kca[b]].cpu(), ref)


@pytest.mark.parametrize("name,B,L,NL,sthr", [("fpn_ssd_640_b64", 64, 0, 21, 0.3), ("retinaface_640_b128", 128, 5, 2, 0.5)])
def test_prior_box_configs(name, B, L, NL, sthr):
    """BASELINE configs[1] (FPN-SSD 640, batch 64, 20 classes) and configs[2] (RetinaFace 640, batch 128,
    5-point landmarks): prior-box decode + class-aware NMS."""
    P, ithr = 50400, 0.4
    priors = O.make_priors(640)
    assert priors.shape == (P, 4)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, P, 4 + 2 * L + NL, generator=g, device=DEV)
    x[..., 4 + 2 * L + 1:] = x[..., 4 + 2 * L + 1:] * 2 - 3
    pp = V.PriorBoxPostProcess(priors, n_landmarks=L).to(DEV)
    st, it = torch.tensor([sthr]), torch.tensor([ithr])
    res = pp(x, st, it)
    cand = ops.prior_decode(x, priors.to(DEV), L, sthr)
    counts = cand.counts.cpu().tolist()
    assert len(res) == B and all(r.shape[1] == 6 + 2 * L for r in res)
    sample = [0, B // 2, B - 1]
    for b in sample:
        bb, sc, ci, det = O.prior_box_decode(x[b:b + 1].cpu(), priors, st, L)
        k = counts[b]
        assert k == bb.shape[0]
        assert torch.allclose(cand.boxes[b, :k].cpu(), bb, rtol=1e-5, atol=1e-6)
        assert torch.allclose(cand.scores[b, :k].cpu(), sc, rtol=1e-5, atol=1e-6)
        assert torch.equal(cand.cls[b, :k].cpu(), ci)
        assert torch.allclose(cand.det[b, :k].cpu(), det[0], rtol=1e-5, atol=1e-6)
        gb, gs, gc = cand.boxes[b, :k].cpu(), cand.scores[b, :k].cpu(), cand.cls[b, :k].cpu()
        keep = O.batched_nms(gb, gs, gc, it)              # bit-exact keep from identical candidates
        assert torch.equal(res[b].cpu(), cand.det[b, :k].cpu()[keep])
    for b in range(0, B, 9):
        k = counts[b]
        keep = V.batched_nms(cand.boxes[b, :k], cand.scores[b, :k], cand.cls[b, :k], it)
        assert torch.equal(cand.det[b, :k][keep], res[b])
        check_kept_properties(res[b], cand.scores[b, :k][keep], cand.cls[b, :k][keep], ithr)


def test_export_check_thresholds_416():
    """The reference's export-check run (scripts/export_check/onnx_check.py:280-285): random uint8 image,
    score_threshold 0.0, iou_threshold 1.0 -> every row survives, nothing is suppressed."""
    S, C = 416, 80
    pred = O.yolo_head_eval(O.synth_yolo_heads(1, S, C, seed=0), S)
    st, it = torch.tensor([0.0]), torch.tensor([1.0])
    res = V.YoloV3PostProcess(img_size=S)(pred.to(DEV), st, it)
    ref = O.yolo_postprocess(pred, st, it, S)
    assert res[0].shape == (10647, 6)
    assert torch.equal(res[0].cpu(), ref[0])
    img = (np.random.RandomState(0).rand(1, S, S, 3) * 255).astype(np.uint8)
    x = V.Normalize()(torch.from_numpy(img).to(DEV))
    assert torch.equal(x.cpu(), O.normalize(torch.from_numpy(img), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]))
