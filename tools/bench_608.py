#!/usr/bin/env python3
"""YOLOv3-608 (the shipped experiments/configs/yolov3_darknet_608.yml size), 80 classes, pre-process + fused
post-process, pipelined depth 2.  usage: bench_608.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vortex_amd as V  # noqa: E402

dev = torch.device("cuda:0")
S, C = 608, 80
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device=dev).manual_seed(5)
img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8, device=dev)
heads = []
for G in (S // 32, S // 16, S // 8):
    t = torch.randn(B, 3, 5 + C, G, G, generator=g, device=dev)
    t[:, :, 2:4] *= 0.5
    t[:, :, 4] = t[:, :, 4] * 2 - 4
    t[:, :, 5:] = t[:, :, 5:] * 1.5 - 2
    heads.append(t.view(B, 3 * (5 + C), G, G).contiguous())
norm = V.Normalize([0.0] * 3, [1.0] * 3).to(dev)           # yolov3_darknet_608.yml:10-13
post = V.YoloV3FusedPostProcess(S)
st, it = torch.tensor([0.9]), torch.tensor([0.2])


def run(n):
    prev = None
    for _ in range(n):
        norm(img)
        cur = post.forward_async(heads, st, it)
        if prev is not None:
            prev.result()
        prev = cur
    return prev.result()


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = run(50)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
algo = B * (15 * S * S + 3 * sum((S // d) ** 2 for d in (32, 16, 8)) * (5 + C) * 4)
print("YOLOv3-608 b%d: %.4f ms/step  %.0f images/s  %.0f GB/s algorithmic end to end  (%d detections)" % (
    B, dt * 1e3, B / dt, algo / dt / 1e9, sum(int(r.shape[0]) for r in out)))
