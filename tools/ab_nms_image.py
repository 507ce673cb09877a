#!/usr/bin/env python3
"""A/B of vx_batched_nms between library builds on the dense-YOLO candidates (score_thr 0.06, b256):
   python tools/ab_nms_image.py lib_a.so lib_b.so ...   (candidates come from the in-tree library)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
B = 256
_, heads = bench.synth_on_device(B, dev, seed=4321)
c = ops.yolo_fused_decode(heads, 416, 0.06)
K, D = c.R, c.D
keep_idx = torch.empty((B, K), dtype=torch.int64, device=dev)
out_det = torch.empty((B, K, D), device=dev)
ref = None
for path in sys.argv[1:] or [L.LIB_PATH]:
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.vx_batched_nms_workspace_bytes.restype = ctypes.c_size_t
    lib.vx_batched_nms_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    nb = lib.vx_batched_nms_workspace_bytes(B, K)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    vp, ci, cf, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    lib.vx_batched_nms.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp, vp, vp, ci, vp, vp, cs, vp]
    lib.vx_batched_nms.restype = ci

    def call():
        rc = lib.vx_batched_nms(c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(), c.det.data_ptr(),
                                c.counts.data_ptr(), B, K, D, 80, 0.2, 0, keep_idx.data_ptr(), out_det.data_ptr(),
                                c.keep_counts.data_ptr(), K, c.status.data_ptr(), ws.data_ptr(), nb, L.stream_ptr(dev))
        assert rc == 0, rc
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    sig = (int(c.keep_counts.sum()), int(keep_idx[0, :int(c.keep_counts[0])].sum()))
    ref = ref or sig
    print("%-40s %.4f ms/call  kept %s %s" % (os.path.basename(path), e0.elapsed_time(e1) / 20, sig, "" if sig == ref else "MISMATCH"))
