"""Which pairs of torch's pooled HIP streams overlap (vx_streams_overlap)?  python tools/probe_streams.py [n]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vortex_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
ss = [torch.cuda.Stream(dev) for _ in range(n)]
print("streams", [hex(s.cuda_stream) for s in ss])
for i in range(n):
    row = []
    for j in range(n):
        if i == j:
            row.append(" . ")
            continue
        ov, a, b = ctypes.c_int(0), ctypes.c_float(0), ctypes.c_float(0)
        L.check(L.lib().vx_streams_overlap(ss[i].cuda_stream, ss[j].cuda_stream, ctypes.byref(ov), ctypes.byref(a), ctypes.byref(b)), "probe")
        row.append("%.2f" % (b.value / a.value))
    print(i, " ".join(row), flush=True)
