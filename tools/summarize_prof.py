#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/prof_r*.sh (gpurun_out/prof) into the committed summaries
under profiles/:  <tag>_kernel_stats.csv (the --kernel-trace --stats summary, library kernels first),
<tag>_bench.json (the bench line of that same profiled command) and traffic_<round>.json (HBM bytes per
launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes).

PMC correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None      # newest: gpurun merges every call's files into gpurun_out/


def kname_full(full):
    """'void vx_k_resize_fast<true>(vx_resize_params, ...)' -> 'vx_k_resize_fast<true>': the instantiated kernel, i.e.
    the name up to the parenthesis that opens the argument list (template arguments kept: two instantiations of one
    template are two kernels with their own byte counts and durations)."""
    n = full.strip().strip('"')
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i].strip()
    return n


def kname(full):
    """the template's name without its arguments"""
    return kname_full(full).split("<")[0]


def pmc(path, counter):
    agg = collections.defaultdict(list)
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[kname_full(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in agg.items()}
    # a template with ONE instantiation in the run is also reachable by its bare name (the bench looks kernels up by
    # that); with several instantiations there is no such alias: their averages must not be pooled
    bare = collections.defaultdict(list)
    for k in out:
        bare[k.split("<")[0]].append(k)
    for b, ks in bare.items():
        if len(ks) == 1 and b not in out:
            out[b] = out[ks[0]]
    return out


def main(src, tag, rnd, workload, dominant=None):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = one(os.path.join(src, "stats", "**", "*_kernel_stats.csv"))
    if ks:
        rows = list(csv.reader(open(ks)))
        head, body = rows[0], rows[1:]
        body.sort(key=lambda r: (not kname(r[0]).startswith("vx_"), -float(r[2])))
        with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            csv.writer(f).writerows([head] + body)
    b = os.path.join(src, "bench_stats.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out, "%s_bench.json" % tag))
    fetch = pmc(one(os.path.join(src, "fetch", "**", "*_counter_collection.csv")), "FETCH_SIZE")
    write = pmc(one(os.path.join(src, "write", "**", "*_counter_collection.csv")), "WRITE_SIZE")
    tpath = os.path.join(out, "traffic_%s.json" % rnd)
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("vx_"):
            continue
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB_avg": fk, "WRITE_SIZE_KiB_avg": wk,
                      "hbm_read_bytes_per_launch": int(2 * fk * 1024), "hbm_write_bytes_per_launch": int(wk * 1024)}
    dom = dominant or ("vx_k_yolo_decode_rows" if workload == "unfused" else "vx_k_yolo_fused")
    entry = {"kernels": kernels, "dominant_kernel": dom,
             "correction": "reads = 2 * FETCH_SIZE KiB * 1024 (gfx950 half-count of wide coalesced reads); "
                           "writes = WRITE_SIZE KiB * 1024; separate --pmc passes (tools/prof_%s.sh)" % rnd,
             "command": open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else None}
    if dom in kernels:
        entry["hbm_bytes_per_launch"] = kernels[dom]["hbm_read_bytes_per_launch"] + kernels[dom]["hbm_write_bytes_per_launch"]
    traffic[workload] = entry
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01_default"
    rnd = sys.argv[3] if len(sys.argv) > 3 else "r01"
    workload = sys.argv[4] if len(sys.argv) > 4 else "default"
    main(src, tag, rnd, workload, sys.argv[5] if len(sys.argv) > 5 else None)
