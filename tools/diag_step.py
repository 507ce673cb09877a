#!/usr/bin/env python3
"""Where does a bench step spend its time?  (diagnostic, run on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402
from vortex_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 256
img, heads = bench.synth_on_device(B, dev, 1234)
norm = V.Normalize().to(dev)
post = V.YoloV3FusedPostProcess(img_size=416)
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
sthr, ithr = torch.tensor([thr]), torch.tensor([0.2])


def timeit(fn, n=50, sync_each=False):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_enq / n * 1e3


def full():
    x = norm(img)
    packed, meta = post.forward_device(heads, sthr, ithr)
    return V.postprocess.finish_results(packed, meta)


def dev_only():
    x = norm(img)
    return post.forward_device(heads, sthr, ithr)


def pre_only():
    return norm(img)


def decode_only():
    return ops.yolo_fused_decode(heads, 416, thr)


c = ops.yolo_fused_decode(heads, 416, thr)


def nms_only():
    return ops.candidates_nms(c, 0.2, 80)


print("thr", thr, "candidates", int(c.counts.sum()))
for name, fn, se in (("full step (sync API)", full, False), ("device only, async", dev_only, False),
                     ("device only, sync each", dev_only, True), ("preprocess only", pre_only, False),
                     ("fused decode only", decode_only, False), ("nms only", nms_only, False)):
    tot, enq = timeit(fn, sync_each=se)
    print("%-28s %.4f ms/step   (host enqueue %.4f ms)" % (name, tot, enq))

# pipelined: finish step i-1 after launching step i
for _ in range(3):
    full()
torch.cuda.synchronize()
t0 = time.perf_counter()
prev = None
n = 50
for i in range(n):
    x = norm(img)
    cur = post.forward_device(heads, sthr, ithr)
    cur_meta_host = cur[1].to("cpu", non_blocking=True)
    if prev is not None:
        V.postprocess.finish_results(*prev)
    prev = cur
V.postprocess.finish_results(*prev)
torch.cuda.synchronize()
print("pipelined depth 2             %.4f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
