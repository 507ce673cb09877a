#!/usr/bin/env python3
"""RetinaFace / FPN-SSD pipelines at realistic candidate counts (a handful per image): which fixed costs remain."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_configs as BC  # noqa: E402

if __name__ == "__main__":
    BC.prior_cfg("retinaface_640_b128_l5_sparse(thr 0.9995)", 128, 5, 2, 0.9995)
    BC.prior_cfg("fpn_ssd_640_b64_c20_sparse(thr 0.995)", 64, 0, 21, 0.995)
