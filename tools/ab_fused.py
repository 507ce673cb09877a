#!/usr/bin/env python3
"""Time the fused decode kernel alone (development A/B).  usage: ab_fused.py [score_thr] [sparse]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
img, heads = bench.synth_on_device(256, dev, 1234)
for _ in range(5):
    c = ops.yolo_fused_decode(heads, 416, thr, threshold_first=sparse)
torch.cuda.synchronize()
L.profile_enable(200)
for _ in range(100):
    ops.yolo_fused_decode(heads, 416, thr, threshold_first=sparse)
torch.cuda.synchronize()
k = L.profile_read()["vx_k_yolo_fused"]
gb = 256 * 3619980 / 1e9
print("%s thr %g (%d candidates) env %s: avg %.4f ms  min %.4f ms  -> %.0f GB/s algorithmic" % (
    "threshold-first" if sparse else "streaming", thr, int(c.counts.sum()), os.environ.get("VX_SPARSE_MAX", "-"),
    k["avg_ms"], k["min_ms"], gb / k["avg_ms"] * 1e3))
