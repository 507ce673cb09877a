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
from .postprocess import PendingDetections, _f, mirror_is_coherent


class _Captured:
    """One captured chain with its static buffers (packed rows, meta words, pinned mirror, pre-process output)."""
    __slots__ = ("graph", "preprocessed", "packed", "meta", "host", "pending")


class GraphedPostProcess:
    """graph = GraphedPostProcess(post, inputs, score_threshold, iou_threshold[, preprocess, images]);
    fill `inputs` (and `images`) in place, then `graph()` -> tuple of per-image detections (fresh tensors),
    or `graph.launch()` -> PendingDetections to overlap the wait with other work.  `graph.preprocessed` is the
    static output of the last launched pre-process (input of the backbone).

    `depth` launches may be in flight: the chain is captured `depth` times, each capture with its own static output
    buffers and pinned mirror, used round-robin.  A launch that comes back to a buffer set whose previous result has
    not been collected yet first waits for that launch and copies its rows out, so an uncollected PendingDetections
    never changes under its owner."""

    def __init__(self, post, inputs: Union[torch.Tensor, Sequence[torch.Tensor]], score_threshold, iou_threshold,
                 preprocess=None, images: Optional[torch.Tensor] = None, warmup: int = 3, depth: int = 2):
        if not hasattr(post, "forward_unpacked"):
            raise RuntimeError("GraphedPostProcess needs a post-process with forward_unpacked (BatchedNMSPostProcess family)")
        self.post, self.inputs, self.images, self.preprocess = post, inputs, images, preprocess
        self._sthr, self._ithr = _f(score_threshold), _f(iou_threshold)
        first = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        self.device = first.device
        self.preprocessed = None
        # Decided BEFORE warm-up and capture: the check allocates and waits on an event, neither of which may happen
        # inside a capture (the first mirror_is_coherent call of a process must therefore be outside any user capture
        # too).  When kernel stores to pinned memory are not host-visible, the chain captures a D2H copy of the
        # meta words instead of the kernel-side mirror -- the same fallback the eager path takes.
        self._coherent = mirror_is_coherent(self.device)
        self._n = 0
        self._sets = []
        side = torch.cuda.Stream(self.device)
        for i in range(max(int(depth), 1)):
            c = _Captured()
            # pinned mirror of the counts/status words: allocated before the capture (no host allocation inside it)
            c.host = torch.empty((2 * first.size(0) + 1,), dtype=torch.int32, pin_memory=True)
            c.pending = None
            if i == 0:
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):               # warm-up off the capture (allocator pools, lazy init)
                    for _ in range(max(warmup, 1)):
                        self._chain(c.host)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
            c.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(c.graph):
                c.preprocessed, c.packed, c.meta = self._chain(c.host)
            self._sets.append(c)
        self.preprocessed = self._sets[0].preprocessed

    def _chain(self, host):
        x = self.preprocess(self.images) if (self.preprocess is not None and self.images is not None) else None
        out_det, keep_counts, meta = self.post.forward_unpacked(self.inputs, self._sthr, self._ithr)
        if self._coherent:
            packed = ops.pack_rows(out_det, keep_counts, meta, host)
        else:
            packed = ops.pack_rows(out_det, keep_counts)
            host.copy_(meta, non_blocking=True)
        return x, packed, meta

    def launch(self) -> PendingDetections:
        c = self._sets[self._n % len(self._sets)]
        self._n += 1
        if c.pending is not None:
            c.pending.detach()                              # its rows leave the static buffer before the replay rewrites it
            if c.pending._copied is not None:               # (stream order: the replay runs after that copy)
                torch.cuda.current_stream(self.device).wait_event(c.pending._copied)
        c.graph.replay()
        self.preprocessed = c.preprocessed
        c.pending = _GraphPending(c.packed, c.meta, c.host)
        return c.pending

    def __call__(self) -> Tuple[torch.Tensor, ...]:
        return self.launch().result()


class _GraphPending(PendingDetections):
    """The packed rows live in a static buffer of the graph: result() hands out copies, and detach() -- called by
    the owner graph before it replays into the same buffers -- takes them early, so a later replay cannot change
    detections somebody still holds or has yet to collect."""
    _rows = None
    _counts = None
    _copied = None

    def detach(self):
        if self._counts is None:
            self._event.synchronize()
            m = self._host.tolist()
            B = (len(m) - 1) // 2
            self._status, self._counts = int(m[-1]), m[B:2 * B]
            self._rows = self.packed[:sum(self._counts)].clone()
            # no host wait: the copy is ordered on its stream, and a replay into the same buffers waits for this event
            self._copied = torch.cuda.Event()
            self._copied.record(torch.cuda.current_stream(self._rows.device))

    def result(self):
        self.detach()
        ops.raise_on_status(self._status)
        return tuple(self._rows.unsafe_split_with_sizes(self._counts))
