"""DETR post-process behind the reference's module API (networks/models/detection/detr.py:531-559)."""
import torch
import torch.nn as nn

from . import ops
from .postprocess import PendingDetections, _f, finish_results


class DETRPostProcess(nn.Module):
    """Same constructor, `additional_inputs`, forward signature, error and return type (tuple of
    f32[K_i, 6] = x1,y1,x2,y2,conf,label in query order) as the reference; all images in one launch."""

    def __init__(self, **kwargs):
        super().__init__()
        self.additional_inputs = (
            ('score_threshold', (1,)),
        )

    def forward_device(self, inputs: torch.Tensor, score_threshold):
        if inputs.ndim != 3:
            raise RuntimeError("DETR postprocess input must have dimension of 3 "
                               "(num_batch, num_queries, (num_classes + 1) + 4)")
        if inputs.size(0) == 0:
            raise RuntimeError("DETR postprocess needs at least one image")
        rows, meta = ops.detr_decode(inputs, _f(score_threshold))
        B = inputs.size(0)
        return ops.pack_rows(rows, meta[B:2 * B]), meta

    def forward_async(self, inputs: torch.Tensor, score_threshold, **kwargs) -> PendingDetections:
        if inputs.ndim != 3:
            raise RuntimeError("DETR postprocess input must have dimension of 3 "
                               "(num_batch, num_queries, (num_classes + 1) + 4)")
        rows, meta = ops.detr_decode(inputs, _f(score_threshold))
        B = inputs.size(0)
        return PendingDetections.packing(rows, meta[B:2 * B], meta)

    @torch.no_grad()
    def forward(self, inputs: torch.Tensor, score_threshold: torch.Tensor, **kwargs) -> torch.Tensor:
        if inputs.ndim == 3 and inputs.size(0) == 0:
            return tuple()                               # the reference's loop over zero images (detr.py:545-548)
        return self.forward_async(inputs, score_threshold).result()
