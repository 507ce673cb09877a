"""GPU tests of the exporter-compatibility layer: custom ops == module path, traceable by torch.jit.trace."""
import pytest
import torch

import vortex_amd as V
from vortex_amd import export
from oracle import vortex_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_custom_ops_match_oracle():
    boxes, scores, cls = O.synth_stress_boxes(3000, 7, 21)
    k = torch.ops.vortex_amd.batched_nms(boxes.to(DEV), scores.to(DEV), cls.to(DEV), 0.45)
    assert torch.equal(k.cpu(), O.batched_nms(boxes, scores, cls, torch.tensor([0.45])))
    k = torch.ops.vortex_amd.nms(boxes.to(DEV), scores.to(DEV), 0.3)
    assert torch.equal(k.cpu(), O.nms(boxes, scores, 0.3))
    img = O.synth_images(2, 32, seed=3)
    y = torch.ops.vortex_amd.normalize(img.to(DEV), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], 255, False)
    assert torch.equal(y.cpu(), O.normalize(img, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]).contiguous())


def test_trace_records_the_op_and_hook_matches_module_path():
    boxes, scores, cls = O.synth_stress_boxes(500, 3, 22)
    det = torch.cat((boxes, scores[:, None], cls[:, None].float()), 1)[None]

    class M(torch.nn.Module):
        def forward(self, b, s):
            return torch.ops.vortex_amd.nms(b, s, 0.4)

    traced = torch.jit.trace(M(), (boxes.to(DEV), scores.to(DEV)), check_trace=False)
    assert "vortex_amd::nms" in str(traced.graph)
    assert torch.equal(traced(boxes.to(DEV), scores.to(DEV)).cpu(), O.nms(boxes, scores, 0.4))

    nms_mod = V.BatchedNMS()
    ref = nms_mod(detections=det.to(DEV), class_indexes=cls.to(DEV), bboxes=boxes.to(DEV), scores=scores.to(DEV),
                  iou_threshold=torch.tensor([0.4]))
    post = V.YoloV3PostProcess(img_size=64)
    export.use_custom_ops(post)
    got = post.nms(detections=det.to(DEV), class_indexes=cls.to(DEV), bboxes=boxes.to(DEV), scores=scores.to(DEV),
                   iou_threshold=torch.tensor([0.4]))
    assert torch.equal(got, ref)


def test_tracing_the_postprocess_after_use_custom_ops_records_the_op():
    """The reference's TorchScript exporter traces the whole predictor after swapping nms_fn
    (exporter/torchscript.py:35-45): with the hook set, forward() runs the per-image composition through the
    registered op, so the traced graph holds vortex_amd::batched_nms -- and the results equal the batched path's."""
    S, C = 64, 3
    pred = O.yolo_head_eval(O.synth_yolo_heads(1, S, C, seed=31), S).to(DEV)
    st, it = torch.tensor([0.06]), torch.tensor([0.3])
    post = V.YoloV3PostProcess(img_size=S)
    ref = post(pred, st, it)
    export.use_custom_ops(post)
    got = post(pred, st, it)
    assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.post = post

        def forward(self, x):
            return self.post(x, st, it)[0]

    traced = torch.jit.trace(M(), (pred,), check_trace=False)
    assert "vortex_amd::batched_nms" in str(traced.inlined_graph)
