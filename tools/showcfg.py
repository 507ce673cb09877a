#!/usr/bin/env python3
"""Development aid: one line per record of a tools/bench_configs.py JSONL file."""
import json, sys
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except Exception:
        continue
    print(d.get('config'), d.get('images_per_s'), d.get('ms_per_step'), d.get('kernels_ms'), 'det/img', d.get('detections_per_image'),
          d.get('kept_per_image'), 'tests', (d.get('nms') or {}).get('iou_tests_per_step'), (d.get('spread') or {}).get('min_ms_per_step'), (d.get('spread') or {}).get('max_ms_per_step'), d.get('error') or '')
