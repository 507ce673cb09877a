#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload (DESIGN.md §5 note; never bench.py's `value`): the uint8 images
start in pinned HOST memory and are copied to the GPU every step (double-buffered on a copy stream, overlapped
with the previous step's kernels); the raw heads stay device-resident because in deployment they are produced
by the backbone on the GPU.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402

dev = torch.device("cuda:0")
B, steps, warmup = 256, 30, 5
img_dev, heads = bench.synth_on_device(B, dev, 1234)
host = [img_dev.cpu().pin_memory() for _ in range(2)]
stage = [torch.empty_like(img_dev) for _ in range(2)]
norm = V.Normalize().to(dev)
post = V.YoloV3FusedPostProcess(img_size=416)
sthr, ithr = torch.tensor([0.9]), torch.tensor([0.2])
copy_stream = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
copied = [torch.cuda.Event() for _ in range(2)]
consumed = [torch.cuda.Event() for _ in range(2)]


def run(n):
    prev = None
    for i in range(n):
        s = i & 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[s])
            stage[s].copy_(host[s], non_blocking=True)
            copied[s].record(copy_stream)
        main.wait_event(copied[s])
        x = norm(stage[s])
        consumed[s].record(main)
        cur = post.forward_async(heads, sthr, ithr)
        if prev is not None:
            prev.result()
        prev = cur
    return x, prev.result()


for e in consumed:
    e.record(main)
run(warmup)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
h2d = B * 416 * 416 * 3
print(json.dumps({"what": "bench workload with the uint8 images copied from pinned host memory every step",
                  "images_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
                  "h2d_bytes_per_step": h2d, "h2d_GBps": round(h2d * steps / dt / 1e9, 2)}))
