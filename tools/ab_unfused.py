#!/usr/bin/env python3
"""Time K2 (three YoloV3Layer launches) and K3 (decode rows) of the API-boundary path (development A/B)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
B, S, C = 256, 416, 80
img, heads = bench.synth_on_device(B, dev, 1234)
layers, off = [], 0
for h, mask in zip(heads, ((6, 7, 8), (3, 4, 5), (0, 1, 2))):
    G = h.shape[-1]
    layers.append((V.YoloV3Layer(S, (G, G), list(mask), ops.COCO_ANCHORS, C).to(dev).eval(), off))
    off += 3 * G * G
decoded = torch.empty((B, off, 5 + C), device=dev)


def step():
    for (layer, o), h in zip(layers, heads):
        layer(h, out=decoded, row_offset=o)
    return ops.yolo_decode(decoded, S, 0.9)


for _ in range(5):
    step()
torch.cuda.synchronize()
L.profile_enable(400)
for _ in range(50):
    step()
torch.cuda.synchronize()
k = L.profile_read()
gb = B * off * (5 + C) * 4 / 1e9
k2, k3 = k["vx_k_yolo_head_decode"], k["vx_k_yolo_decode_rows"]
print("K2_TC=%s K3=%s: K2 3 launches %.4f ms/step -> %.0f GB/s (r+w)   K3 %.4f ms -> %.0f GB/s" % (
    os.environ.get("VX_K2_TC", "64"), os.environ.get("VX_K3", "-"), k2["total_ms"] / 50, 2 * gb / (k2["total_ms"] / 50) * 1e3,
    k3["avg_ms"], gb / k3["avg_ms"] * 1e3))
