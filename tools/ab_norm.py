#!/usr/bin/env python3
"""K1 alone at the BASELINE image sizes: kernel time and fraction of 8 TB/s (3 B read + 12 B written per pixel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vortex_amd as V  # noqa: E402

dev = torch.device("cuda:0")
norm = V.Normalize([0.5] * 3, [0.5] * 3).to(dev)
for B, S in ((256, 416), (128, 640), (64, 640), (64, 512), (32, 608)):
    img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device=dev)
    for _ in range(5):
        norm(img)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        norm(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    by = B * S * S * 15
    print("b%-4d %4d^2  %.4f ms  %.0f GB/s  %.3f of peak" % (B, S, ms, by / ms / 1e6, by / ms / 1e6 / 8000))
