#!/usr/bin/env python3
"""profiles/r03_* and profiles/traffic_r03.json from the gpurun_out/prof_<tag> directories of tools/prof_r03.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_prof as SP  # noqa: E402

ROOT = SP.ROOT
# tag -> (workload key in traffic_r03.json, dominant kernel)
TAGS = {
    "default": ("default", "vx_k_yolo_fused"),
    "dense": ("dense", "vx_k_yolo_fused"),
    "unfused": ("unfused", "vx_k_yolo_decode_rows"),
    "retinaface": ("retinaface_640_b128", "vx_k_prior_decode_reg"),
    "fpn_ssd": ("fpn_ssd_640_b64", "vx_k_prior_decode"),
    "stress": ("nms_stress_100k_b32_top300", "vx_k_nms_image"),
    "stress_all": ("nms_stress_100k_b32_all", "vx_k_bm_pairs"),
    "resize720p": ("yolo416_b256_resize720p", "vx_k_resize_fast<true>"),
    "dense_cfg": ("yolo416_b256_dense", "vx_k_yolo_fused"),
    "b1": ("yolo416_b1_latency", "vx_k_yolo_fused"),
}
for tag, (workload, dom) in TAGS.items():
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    if os.path.isdir(src):
        print("==", tag)
        SP.main(src, "r03_" + tag, "r03", workload, dom)
