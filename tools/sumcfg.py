import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"], d.get("images_per_s"), d.get("ms_per_step"), {k: v for k, v in d.get("kernels_ms", {}).items() if "nms" in k or "sort" in k or "prior" in k}, (d.get("nms") or {}).get("iou_tests_per_step"), d.get("error"))
