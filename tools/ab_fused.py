#!/usr/bin/env python3
"""Time the fused decode kernel alone (development A/B).  usage: ab_fused.py [score_thr] [sparse|stream] [S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 416
B, C = int(os.environ.get('VX_AB_B', '256')), 80
g = torch.Generator(device=dev).manual_seed(1234)
heads = []
for G in (S // 32, S // 16, S // 8):
    t = torch.randn(B, 3, 5 + C, G, G, generator=g, device=dev)
    t[:, :, 2:4] *= 0.5
    t[:, :, 4] = t[:, :, 4] * 2 - 4
    t[:, :, 5:] = t[:, :, 5:] * 1.5 - 2
    heads.append(t.view(B, 3 * (5 + C), G, G).contiguous())
for _ in range(5):
    c = ops.yolo_fused_decode(heads, S, thr, threshold_first=sparse)
torch.cuda.synchronize()
L.profile_enable(200)
for _ in range(100):
    ops.yolo_fused_decode(heads, S, thr, threshold_first=sparse)
torch.cuda.synchronize()
k = L.profile_read()["vx_k_yolo_fused"]
gb = B * ops.yolo_rows(S) * (5 + C) * 4 / 1e9
print("B=%d " % B + "%s S=%d thr %g (%d candidates) variant %s: avg %.4f ms  min %.4f ms  -> %.0f GB/s algorithmic" % (
    "threshold-first" if sparse else "streaming", S, thr, int(c.counts.sum()), os.environ.get("VX_FUSED_VARIANT", "-"),
    k["avg_ms"], k["min_ms"], gb / k["avg_ms"] * 1e3))
