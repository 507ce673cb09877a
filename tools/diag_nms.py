#!/usr/bin/env python3
"""Phase breakdown of vx_k_nms_image from its in-kernel clock stamps (diagnostic, GPU box)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.06
B = 256
img, heads = bench.synth_on_device(B, dev, 1234)
c = ops.yolo_fused_decode(heads, 416, thr)
lib = L.lib()
K = c.R
nb = lib.vx_batched_nms_workspace_bytes(B, K)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
keep_idx = torch.empty((B, K), dtype=torch.int64, device=dev)
out_det = torch.empty((B, K, 6), device=dev)
L.profile_enable(64)                                  # the kernel writes its phase clocks only while profiling
for it in range(3):
    rc = lib.vx_batched_nms(c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(), c.det.data_ptr(),
                            c.counts.data_ptr(), B, K, 6, 80, 0.2, 0, keep_idx.data_ptr(), out_det.data_ptr(),
                            c.keep_counts.data_ptr(), K, c.status.data_ptr(), ws.data_ptr(), nb,
                            L.stream_ptr(dev))
    assert rc == 0
torch.cuda.synchronize()
off = lib.vx_batched_nms_stamps_offset(B, K)
st = ws[off:off + B * 136 * 8].view(torch.int64).view(B, 136)[:, :8].cpu().double()
d = (st[:, 1:] - st[:, :-1])
names = ["keys", "sort1", "gather", "sweep-wave", "sweep-block", "load+sort2", "output"]
print("candidates/img %.0f kept/img %.0f" % (c.counts.float().mean(), c.keep_counts.float().mean()))
tot = (st[:, 7] - st[:, 0])
print("total cycles: mean %.0f max %.0f  (100 MHz ticks if s_memrealtime, shader clocks if s_memtime)" % (tot.mean(), tot.max()))
for i, n in enumerate(names):
    print("%-12s mean %8.0f  max %8.0f  share %.1f%%" % (n, d[:, i].mean(), d[:, i].max(), 100 * d[:, i].mean() / tot.mean()))
