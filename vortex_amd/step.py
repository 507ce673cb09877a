"""One library call per batch for the two halves of a YOLOv3 predictor that sit around the backbone
(reference predictor/base_module.py:98-102: preprocess -> model -> postprocess).

`YoloV3PipelineStep(preprocess, postprocess)` takes this batch's frames AND the head tensors the backbone produced,
and issues `vx_yolo_pipeline_step`: pre-process + fused decode + NMS + packing chained on the current stream by ONE
marshalled call.  Its buffers (normalised batch, packed rows, counts, pinned mirror, workspace) are allocated once per
input shape and reused from a small ring, so a synchronous batch-1 caller pays neither two library calls nor five
tensor allocations per image.  Results are identical to `postprocess(heads, ...)` / `preprocess(images)`.

Lifetime: the tensors returned by call k are views of ring slot k % depth and are overwritten by call k + depth
(like the static buffers of graphs.GraphedPostProcess); copy what must live longer."""
from typing import Sequence, Tuple

import torch

from . import _lib as L, ops
from .postprocess import YoloV3FusedPostProcess, _split_results, _f, mirror_is_coherent
from .preprocess import Normalize


class _Slot:
    __slots__ = ("norm", "packed", "meta", "host", "ws", "nb", "event", "busy")


class YoloV3PipelineStep:
    def __init__(self, preprocess: Normalize, postprocess: YoloV3FusedPostProcess, depth: int = 3,
                 letterbox: bool = False, pad_value: int = 0):
        assert isinstance(preprocess, Normalize) and isinstance(postprocess, YoloV3FusedPostProcess)
        assert postprocess.tail_stream is None, "the one-call step keeps everything on the caller's stream"
        self.pre, self.post, self.depth = preprocess, postprocess, max(1, int(depth))
        self.resize_mode = L.VX_RESIZE_LETTERBOX if letterbox else L.VX_RESIZE_STRETCH
        self.pad_value = int(pad_value)
        self._rings = {}
        self._k = 0
        d = postprocess.decoder
        self._S = int(d.img_size)
        self._f18 = L.f18([float(v) for a in d.anchors for v in a])
        self._flags = int(bool(d.threshold)) | (2 if d.nan_check == "survivors" else 0)
        self._optimistic = ops.yolo_rows(self._S) > ops.NMS_SINGLE_LAUNCH_MAX
        self._consts = None

    def _slot(self, B: int, dev) -> _Slot:
        key = (B, torch.device(dev).index)
        ring = self._rings.get(key)
        if ring is None:
            ring = self._rings[key] = []
            S, tk = self._S, int(self.post.top_k or 0)
            R = ops.yolo_rows(S)
            out_cap = min(R, tk) if tk > 0 else R
            nb = L.lib().vx_yolo_fused_postprocess_workspace_bytes(B, S, tk)
            for _ in range(self.depth):
                s = _Slot()
                s.norm = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
                s.packed = torch.empty((B * out_cap, 6), dtype=torch.float32, device=dev)
                s.meta = torch.empty((2 * B + 1,), dtype=torch.int32, device=dev)
                s.host = torch.empty((2 * B + 1,), dtype=torch.int32, pin_memory=True)
                s.ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=dev)
                s.nb = nb
                s.event = torch.cuda.Event()
                s.busy = False
                ring.append(s)
        s = ring[self._k % self.depth]
        self._k += 1
        if s.busy:
            s.event.synchronize()                       # the slot's previous step (depth calls ago) has finished
        return s

    def launch(self, images: torch.Tensor, heads: Sequence[torch.Tensor], score_threshold, iou_threshold) -> _Slot:
        """Issue the step; returns the slot (event recorded behind the step's last kernel)."""
        L.require_cuda(images, "images")
        assert images.dtype == torch.uint8 and images.dim() == 4 and images.size(3) == 3 and images.is_contiguous()
        assert len(heads) == 3
        B, Hs, Ws = int(images.size(0)), int(images.size(1)), int(images.size(2))
        S = self._S
        for h, dv in zip(heads, (32, 16, 8)):
            L.require_cuda(h, "head")
            assert h.dtype == torch.float32 and h.dim() == 4 and h.size(0) == B and h.size(2) == S // dv == h.size(3) \
                and h.is_contiguous()
        C = int(heads[0].size(1)) // 3 - 5
        dev = images.device
        if self._consts is None:
            mean, std, scaler = self.pre._consts()
            self._consts = (L.f3(mean), L.f3(std), int(scaler), int(bool(self.pre._flip)))
        mean, std, scaler, flip = self._consts
        s = self._slot(B, dev)
        coherent = mirror_is_coherent(dev)
        flags = self._flags | (4 if self._optimistic else 0)
        rc = L.lib().vx_yolo_pipeline_step(
            images.data_ptr(), s.norm.data_ptr(), Hs, Ws, mean, std, scaler, flip, self.resize_mode, self.pad_value,
            heads[0].data_ptr(), heads[1].data_ptr(), heads[2].data_ptr(), B, S, C, self._f18,
            _f(score_threshold), flags, _f(iou_threshold), int(self.post.top_k or 0), s.packed.data_ptr(),
            s.meta.data_ptr(), s.host.data_ptr() if coherent else None, s.ws.data_ptr(), s.nb, L.stream_ptr(dev))
        L.check(rc, "vx_yolo_pipeline_step")
        if not coherent:
            s.host.copy_(s.meta, non_blocking=True)
        s.event.record(torch.cuda.current_stream(dev))
        s.busy = True
        return s

    def __call__(self, images: torch.Tensor, heads: Sequence[torch.Tensor], score_threshold,
                 iou_threshold) -> Tuple[torch.Tensor, Tuple[torch.Tensor, ...]]:
        """-> (normalised batch f32[B,3,S,S], per-image detections): `preprocess(images)` and
        `postprocess(heads, score_threshold, iou_threshold)` of the wrapped modules."""
        if images.size(0) == 0:
            return torch.empty((0, 3, self._S, self._S), dtype=torch.float32, device=images.device), tuple()
        s = self.launch(images, heads, score_threshold, iou_threshold)
        s.event.synchronize()
        s.busy = False
        m = s.host.tolist()
        if m[-1] & L.VX_STATUS_DEFERRED:                # the optimistic NMS declined an image: the general path decides
            return s.norm, self.post(heads, score_threshold, iou_threshold)
        return s.norm, _split_results(s.packed, m)
