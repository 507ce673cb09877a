"""HIP-graph replay of the post-process (and optionally the pre-process) for fixed shapes.

At small batch the path is launch-bound: five short kernels plus ~10 torch allocations cost ~0.09 ms of host
time per call, more than the kernels themselves (YOLOv3-416 b1: 0.055 ms of GPU work in a 0.14 ms step).  Every
entry point of libvortex_hip takes its stream and neither allocates nor synchronises, so the whole chain can be
captured once with torch.cuda.graph() (a hipGraph on ROCm) and replayed with a single launch.

The captured call reads the SAME input tensors on every replay: the producer (backbone, or `copy_` from a new
batch) writes into them.  Thresholds are kernel arguments and therefore fixed at capture time.  No reference
counterpart: the reference runs eagerly (predictor/base_module.py:98-102)."""
from typing import Optional, Sequence, Tuple, Union

import torch

from . import ops
from .postprocess import PendingDetections, _f


class GraphedPostProcess:
    """graph = GraphedPostProcess(post, inputs, score_threshold, iou_threshold[, preprocess, images]);
    fill `inputs` (and `images`) in place, then `graph()` -> tuple of per-image detections (fresh tensors),
    or `graph.launch()` -> PendingDetections to overlap the wait with other work.  `graph.preprocessed` is the
    static output of the captured pre-process (input of the backbone)."""

    def __init__(self, post, inputs: Union[torch.Tensor, Sequence[torch.Tensor]], score_threshold, iou_threshold,
                 preprocess=None, images: Optional[torch.Tensor] = None, warmup: int = 3):
        if not hasattr(post, "forward_unpacked"):
            raise RuntimeError("GraphedPostProcess needs a post-process with forward_unpacked (BatchedNMSPostProcess family)")
        self.post, self.inputs, self.images, self.preprocess = post, inputs, images, preprocess
        self._sthr, self._ithr = _f(score_threshold), _f(iou_threshold)
        first = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        self.device = first.device
        self.preprocessed = None
        # pinned mirror of the counts/status words: allocated before the capture (no host allocation inside it)
        self._host = torch.empty((2 * first.size(0) + 1,), dtype=torch.int32, pin_memory=True)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                       # warm-up off the capture (allocator pools, lazy init)
            for _ in range(max(warmup, 1)):
                self._chain()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.preprocessed, self._packed, self._meta = self._chain()

    def _chain(self):
        x = self.preprocess(self.images) if (self.preprocess is not None and self.images is not None) else None
        out_det, keep_counts, meta = self.post.forward_unpacked(self.inputs, self._sthr, self._ithr)
        packed = ops.pack_rows(out_det, keep_counts, meta, self._host)
        return x, packed, meta

    def launch(self) -> PendingDetections:
        self.graph.replay()
        return _GraphPending(self._packed, self._meta, self._host)

    def __call__(self) -> Tuple[torch.Tensor, ...]:
        return self.launch().result()


class _GraphPending(PendingDetections):
    """The packed rows live in the graph's static buffer: result() hands out copies, so the next replay cannot
    change detections the caller still holds."""

    def result(self):
        self._event.synchronize()
        m = self._host.tolist()
        B = (len(m) - 1) // 2
        ops.raise_on_status(int(m[-1]))
        kc = m[B:2 * B]
        return tuple(self.packed[:sum(kc)].clone().unsafe_split_with_sizes(kc))
