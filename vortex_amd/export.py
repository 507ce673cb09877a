"""Exporter compatibility: the HIP entry points as `torch.library` custom ops, so graphs captured
by `torch.jit.trace` (exporter/torchscript.py:35-45 traces the predictor), `torch.export` or `torch.compile`
contain one opaque node per call instead of baked-in ctypes side effects, plus the ONNX symbolic the reference
registers for `torchvision::nms` (exporter/onnx.py:42-58) under our op name.

    torch.ops.vortex_amd.nms(boxes, scores, iou_threshold)                 -> int64[K']
    torch.ops.vortex_amd.batched_nms(boxes, scores, idxs, iou_threshold)   -> int64[K']
    torch.ops.vortex_amd.normalize(input_u8_nhwc, mean, std, scaler, flip) -> f32[N,3,H,W]

`use_custom_ops(postprocess)` is the counterpart of the reference's export-time hook
`predictor.postprocess.nms.nms_fn = torch.jit.script(...)` (exporter/torchscript.py:38-41).
The ops run the same HIP kernels (GPU tensors only; a CPU tensor raises VortexHipError)."""
from typing import List

import torch

from . import ops
from . import postprocess as _pp

_LIB = "vortex_amd"


@torch.library.custom_op(_LIB + "::nms", mutates_args=())
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return _pp.nms(boxes, scores, iou_threshold).clone()


@nms.register_fake
def _(boxes, scores, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


@torch.library.custom_op(_LIB + "::batched_nms", mutates_args=())
def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return _pp.batched_nms(boxes, scores, idxs, iou_threshold).clone()


@batched_nms.register_fake
def _(boxes, scores, idxs, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


@torch.library.custom_op(_LIB + "::normalize", mutates_args=())
def normalize(input: torch.Tensor, mean: List[float], std: List[float], scaler: int, flip: bool) -> torch.Tensor:
    return ops.preprocess(input, mean, std, scaler, flip)


@normalize.register_fake
def _(input, mean, std, scaler, flip):
    n, h, w, _ = input.shape
    return input.new_empty((n, 3, h, w), dtype=torch.float32)


def _symbolic_nms(g, boxes, scores, iou_threshold):
    """exporter/onnx.py:51-58 (`symbolic_multi_label_nms`) for our op: ONNX NonMaxSuppression, column 2."""
    import sys
    from torch.onnx.symbolic_helper import _get_const, _unsqueeze_helper, _squeeze_helper
    thr = float(_get_const(iou_threshold, "f", "iou_threshold"))
    boxes = _unsqueeze_helper(g, boxes, [0])
    scores = _unsqueeze_helper(g, _unsqueeze_helper(g, scores, [0]), [0])
    max_output_per_class = g.op("Constant", value_t=torch.tensor([sys.maxsize], dtype=torch.long))
    iou = g.op("Constant", value_t=torch.tensor([thr], dtype=torch.float))
    out = g.op("NonMaxSuppression", boxes, scores, max_output_per_class, iou)
    idx = g.op("Gather", out, g.op("Constant", value_t=torch.tensor([2], dtype=torch.long)), axis_i=1)
    return _squeeze_helper(g, idx, [1])


def register_onnx_symbolics(opsets=(9, 10, 11)) -> bool:
    """Register the NonMaxSuppression symbolic for vortex_amd::nms (TorchScript-based ONNX exporter).
    Returns False when this torch build has no such registry."""
    try:
        from torch.onnx import register_custom_op_symbolic
    except Exception:
        return False
    for v in opsets:
        register_custom_op_symbolic(_LIB + "::nms", _symbolic_nms, v)
    return True


def use_custom_ops(post) -> None:
    """Route a post-process module's NMS through the registered ops (call before tracing / exporting)."""
    if hasattr(post, "nms") and isinstance(post.nms, _pp.BatchedNMS):
        post.nms.nms_fn = torch.ops.vortex_amd.batched_nms
