#!/usr/bin/env python3
"""How much of a multi-stream run had kernels of different streams on the GPU at once?  Reads a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace -d DIR -- python tools/bench_configs.py retinaface --no-cpu`), keeps the launches on streams
other than the process's first one (the StreamRing part of the run) and reports the busy time, the summed kernel time
and the time during which kernels of at least two streams ran.

    python tools/overlap_from_trace.py gpurun_out/prof_ms [out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import Counter


def main(src, out=None):
    f = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    per_stream = Counter(r["Stream_Id"] for r in rows)
    main_stream = min(per_stream, key=lambda s: min(int(r["Start_Timestamp"]) for r in rows if r["Stream_Id"] == s))
    ring = [r for r in rows if r["Stream_Id"] != main_stream and r["Kernel_Name"].startswith(("vx_", "void vx_"))]
    streams = sorted(set(r["Stream_Id"] for r in ring))
    ev = []
    for r in ring:
        ev.append((int(r["Start_Timestamp"]), 1, r["Stream_Id"]))
        ev.append((int(r["End_Timestamp"]), -1, r["Stream_Id"]))
    ev.sort()
    active = Counter()
    last = None
    busy = multi = 0
    for t, d, s in ev:
        if last is not None:
            n = sum(1 for v in active.values() if v > 0)
            if n >= 1:
                busy += t - last
            if n >= 2:
                multi += t - last
        active[s] += d
        last = t
    ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ring)
    by_kernel = Counter()
    for r in ring:
        by_kernel[r["Kernel_Name"].split("(")[0][:40]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rec = {"trace": os.path.relpath(f), "ring_streams": streams, "launches": len(ring),
           "gpu_busy_ms": round(busy / 1e6, 3), "summed_kernel_ms": round(ksum / 1e6, 3),
           "two_or_more_streams_ms": round(multi / 1e6, 3), "frac_of_busy_time_overlapped": round(multi / max(busy, 1), 4),
           "summed_over_busy": round(ksum / max(busy, 1), 3),
           "kernel_ms": {k: round(v / 1e6, 3) for k, v in by_kernel.most_common(8)}}
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
