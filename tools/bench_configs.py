#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line):
FPN-SSD 640 b64, RetinaFace 640 b128, NMS stress 100k x 80 classes b32, YOLOv3-416 b1 (latency).
Prints one JSON line per configuration with per-kernel times from the library's HIP-event hooks."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vortex_amd as V  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")


def run(name, fn, images, steps=20, warmup=3, extra=None):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    L.profile_enable(steps * 40)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = L.profile_read()
    L.profile_disable()
    d = {"config": name, "images_per_s": round(images / dt, 1), "ms_per_step": round(dt * 1e3, 4),
         "kernels_ms_total_per_step": {k: round(v["total_ms"] / steps, 5) for k, v in kern.items()}}
    if extra:
        d.update(extra(out))
    print(json.dumps(d), flush=True)


def make_priors(img_size=640):
    """6 normalised (cx,cy,w,h) priors per cell on strides 8/16/32: P = 50400 at 640 (timing input only)."""
    out = []
    for s in (8, 16, 32):
        g = img_size // s
        ys, xs = torch.meshgrid(torch.arange(g).float(), torch.arange(g).float(), indexing="ij")
        cx, cy = ((xs + 0.5) * s / img_size).reshape(-1, 1), ((ys + 0.5) * s / img_size).reshape(-1, 1)
        per = []
        for k in range(2):
            base = s * 4 * (2 ** (k / 2)) / img_size
            for ar in (1, 2, 3):
                per.append(torch.cat((cx, cy, torch.full_like(cx, base * ar ** 0.5), torch.full_like(cx, base / ar ** 0.5)), 1))
        out.append(torch.stack(per, 1).reshape(-1, 4))
    return torch.cat(out, 0).float().contiguous()


def prior_cfg(name, B, Lm, NL, sthr):
    P = 50400
    priors = make_priors(640).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, P, 4 + 2 * Lm + NL, generator=g, device=dev)
    x[..., 4 + 2 * Lm + 1:] = x[..., 4 + 2 * Lm + 1:] * 2 - 3
    img = torch.randint(0, 256, (B, 640, 640, 3), generator=g, dtype=torch.uint8, device=dev)
    norm = V.Normalize([0.5] * 3, [0.5] * 3).to(dev)
    pp = V.PriorBoxPostProcess(priors, n_landmarks=Lm).to(dev)
    st, it = torch.tensor([sthr]), torch.tensor([0.4])

    def fn():
        norm(img)
        return pp(x, st, it)
    algo = B * (15 * 640 * 640 + P * (4 + 2 * Lm + NL) * 4) + P * 16
    run(name, fn, B, extra=lambda out: {"detections_per_batch": sum(int(r.shape[0]) for r in out),
                                        "algorithmic_bytes_per_step": algo})


def stress_cfg():
    B, K, C = 32, 100_000, 80
    g = torch.Generator(device=dev).manual_seed(9)
    c = torch.rand(B, K, 2, generator=g, device=dev)
    wh = torch.rand(B, K, 2, generator=g, device=dev) * 0.28 + 0.02
    boxes = torch.cat((c - wh / 2, c + wh / 2), 2).contiguous()
    scores = torch.rand(B, K, generator=g, device=dev)
    cls = torch.randint(0, C, (B, K), generator=g, device=dev, dtype=torch.int64)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2).contiguous()

    def fn():
        return ops.batched_nms(boxes, scores, cls, det, None, 0.45, C, 300)
    # NMS is not HBM-bound -> report pair evaluations.  Upper bound = sum over (image, class) of
    # n_c (n_c - 1) / 2 (every pair of a class segment); the greedy sweep skips pairs whose first box is suppressed.
    cnt = torch.zeros(B, C, device=dev).scatter_add_(1, cls, torch.ones_like(cls, dtype=torch.float32))
    pairs = float((cnt * (cnt - 1) / 2).sum())
    run("nms_stress_100k_c80_b32_top300", fn, B,
        extra=lambda out: {"kept_per_image_after_topk": float(out[2].float().mean()),
                           "class_pair_upper_bound_per_step": pairs})


def yolo_b1():
    img, heads = bench.synth_on_device(1, dev, 1)
    norm = V.Normalize().to(dev)
    post = V.YoloV3FusedPostProcess(img_size=416)
    st, it = torch.tensor([0.9]), torch.tensor([0.2])

    def fn():
        norm(img)
        return post(heads, st, it)
    run("yolov3_416_b1_latency", fn, 1, steps=200, warmup=20)
    from vortex_amd.graphs import GraphedPostProcess
    g = GraphedPostProcess(post, heads, st, it, preprocess=norm, images=img)
    run("yolov3_416_b1_latency_hipgraph", lambda: g(), 1, steps=200, warmup=20)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ssd", "retinaface", "stress", "b1"]
    if "ssd" in which:
        prior_cfg("fpn_ssd_640_b64_c20", 64, 0, 21, 0.3)
    if "retinaface" in which:
        prior_cfg("retinaface_640_b128_l5", 128, 5, 2, 0.5)
    if "stress" in which:
        stress_cfg()
    if "b1" in which:
        yolo_b1()
