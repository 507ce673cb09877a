"""Batches in turn on several HIP streams.

The NMS tail of a detection step (one workgroup per image, serial chains) leaves most of the chip idle, and the
HBM-bound kernels in front of it (pre-process, decode) leave the VALUs idle: a serving loop that keeps two batches in
flight gets both busy at once by giving each batch its own stream.  Every entry point of the library takes its
stream, workspace and outputs from the caller ("thread-safety"), so two steps on two streams share
nothing.  Measured on MI355X, three streams with four batches in flight: RetinaFace 640 b128 0.546 -> 0.396 ms per
step, FPN-SSD 640 b64 0.624 -> 0.511 ms; the YOLOv3 steps are HBM-bound end to end and gain nothing."""
import contextlib
import ctypes
from typing import Optional

import torch

from . import _lib as L


def _overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    ov = ctypes.c_int(0)
    L.check(L.lib().vx_streams_overlap(a.cuda_stream, b.cuda_stream, ctypes.byref(ov), None, None), "vx_streams_overlap")
    return bool(ov.value)


def overlapping_stream(other: Optional["torch.cuda.Stream"] = None, tries: int = 8) -> "torch.cuda.Stream":
    """A stream whose kernels run beside those of `other` (default: the current stream) -- probed with
    vx_streams_overlap; the last candidate is returned if none of `tries` does."""
    other = torch.cuda.current_stream() if other is None else other
    cand = None
    with torch.cuda.device(other.device):
        for _ in range(max(1, tries)):
            cand = torch.cuda.Stream(other.device)
            if _overlap(other, cand):
                break
    return cand


class StreamRing:
    """`with ring.next(): pending = postprocess.forward_async(...)` -- the step is enqueued on the ring's next stream,
    which first waits for what the caller's current stream still has in flight (the step's inputs), if anything.  The
    step's tensors are allocated under that stream, so torch's allocator recycles them per stream; `pending.result()`
    waits for the event recorded on it.  Inputs must stay alive until then (they belong to the caller's stream)."""

    _streams = {}        # per device: the streams every ring on it shares (chosen once)

    def __init__(self, n: int = 2, device: Optional[torch.device] = None):
        """HIP spreads a process's streams over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), not in turn:
        of 8 streams drawn from torch's pool on an MI355X, {0,5}, {1,4} and {2,3,7} shared a queue, and two streams on
        one queue run their kernels strictly one after the other.  So the ring picks its streams with the library's
        probe (vx_streams_overlap: a one-wave kernel on each stream, timed) -- each new member must overlap with every
        member already chosen -- once per device; later rings on the device reuse them."""
        assert n >= 1
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        have = StreamRing._streams.setdefault(idx, [])
        spare = []
        with torch.cuda.device(idx):
            for _ in range(16):
                if len(have) >= n:
                    break
                cand = torch.cuda.Stream(dev)
                if all(_overlap(h, cand) for h in have):
                    have.append(cand)
                else:
                    spare.append(cand)
        # `have` holds only streams that overlap pairwise (probed).  More members asked for than the probe found
        # (fewer hardware queues, or a busy GPU that made the probe say no): the rest share a queue with somebody --
        # the results are the same, those steps just run one after the other.
        self.streams = (have + spare)[:int(n)]
        self.overlapping = len(have) >= n
        self._k = 0

    def __len__(self):
        return len(self.streams)

    @contextlib.contextmanager
    def next(self):
        s = self.streams[self._k % len(self.streams)]
        self._k += 1
        cur = torch.cuda.current_stream(s.device)
        if not cur.query():                   # an idle caller stream has produced its inputs already: no event, no
            s.wait_stream(cur)                # barrier packet in front of the step (it costs the overlap: DESIGN §5.0)
        with torch.cuda.stream(s):
            yield s
