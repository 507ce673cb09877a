#!/usr/bin/env python3
"""cProfile of the host side of one pipelined step (development aid): where the Python time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img, heads = bench.synth_on_device(B, dev, 1234)
norm = V.Normalize().to(dev)
post = V.YoloV3FusedPostProcess(img_size=416, nan_check="survivors")
st, it = torch.tensor([0.9]), torch.tensor([0.2])


def run(n):
    prev = None
    for _ in range(n):
        norm(img)
        cur = post.forward_async(heads, st, it)
        if prev is not None:
            prev.result()
        prev = cur
    prev.result()


run(20)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(300)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
