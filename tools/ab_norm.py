#!/usr/bin/env python3
"""Time the normalize kernel alone (development A/B).  usage: ab_norm.py [S] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vortex_amd as V  # noqa: E402
from vortex_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 416
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device=dev)
norm = V.Normalize().to(dev)
for _ in range(5):
    norm(img)
torch.cuda.synchronize()
L.profile_enable(200)
for _ in range(100):
    norm(img)
torch.cuda.synchronize()
k = L.profile_read()["vx_k_normalize/resize"]
gb = B * S * S * 15 / 1e9
print("normalize S=%d B=%d variant %s: avg %.4f ms  min %.4f ms  -> %.0f GB/s avg, %.0f GB/s best" % (
    S, B, os.environ.get("VX_NORM_VARIANT", "-"), k["avg_ms"], k["min_ms"], gb / k["avg_ms"] * 1e3, gb / k["min_ms"] * 1e3))
