#!/usr/bin/env python3
"""Eager vs HIP-graph replay of the bench step (development A/B).  usage: ab_graph.py [B] [threshold_first]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402
from vortex_amd.graphs import GraphedPostProcess  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tf = len(sys.argv) > 2 and sys.argv[2] == "threshold_first"
img, heads = bench.synth_on_device(B, dev, 1234)
norm = V.Normalize().to(dev)
post = V.YoloV3FusedPostProcess(416, nan_check="survivors" if tf else "full")
st, it = torch.tensor([0.9]), torch.tensor([0.2])
g = GraphedPostProcess(post, heads, st, it, preprocess=norm, images=img)


def eager():
    norm(img)
    return post.forward_async(heads, st, it)


def timeit(launch, n=200):
    prev = None
    for _ in range(10):
        launch().result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        cur = launch()
        if prev is not None:
            prev.result()
        prev = cur
    prev.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("eager", eager), ("hipgraph", g.launch)):
    ms = timeit(fn)
    print("B=%d %s %s: %.4f ms/step  %.0f images/s" % (B, "threshold-first" if tf else "full-scan", name, ms, B / ms * 1e3))
