#!/usr/bin/env python3
"""Time the fused decode kernel alone (development A/B; the variant comes from VX_FUSED_VARIANT)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
img, heads = bench.synth_on_device(256, dev, 1234)
for _ in range(5):
    ops.yolo_fused_decode(heads, 416, thr)
torch.cuda.synchronize()
L.profile_enable(200)
for _ in range(100):
    ops.yolo_fused_decode(heads, 416, thr)
torch.cuda.synchronize()
k = L.profile_read()["vx_k_yolo_fused"]
gb = 256 * 3619980 / 1e9
print("variant %s thr %g: avg %.4f ms  min %.4f ms  -> %.0f GB/s avg, %.0f GB/s best" % (
    os.environ.get("VX_FUSED_VARIANT", "0"), thr, k["avg_ms"], k["min_ms"], gb / k["avg_ms"] * 1e3, gb / k["min_ms"] * 1e3))
