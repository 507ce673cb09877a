#!/usr/bin/env python3
"""Time the normalize kernel alone (development A/B; variant from VX_NORM_VARIANT)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402
from vortex_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
img, _ = bench.synth_on_device(256, dev, 1234)
norm = V.Normalize().to(dev)
for _ in range(5):
    norm(img)
torch.cuda.synchronize()
L.profile_enable(200)
for _ in range(100):
    norm(img)
torch.cuda.synchronize()
k = L.profile_read()["vx_k_normalize/resize"]
gb = 256 * 2595840 / 1e9
print("norm variant %s: avg %.4f ms  min %.4f ms  -> %.0f GB/s avg, %.0f GB/s best" % (
    os.environ.get("VX_NORM_VARIANT", "0"), k["avg_ms"], k["min_ms"], gb / k["avg_ms"] * 1e3, gb / k["min_ms"] * 1e3))
