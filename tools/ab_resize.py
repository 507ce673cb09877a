#!/usr/bin/env python3
"""Resize leg of the pre-process alone: 720x1280 uint8 -> 416x416 (fused with normalise, and u8 -> u8), batch 256,
against SURVEY §8d's algorithmic bytes (3*Hs*Ws read + 12*S*S written = 4 841 472 B per image for the fused form)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
B, Hs, Ws, S = 256, 720, 1280, 416
g = torch.Generator(device=dev).manual_seed(3)
img = torch.randint(0, 256, (B, Hs, Ws, 3), generator=g, dtype=torch.uint8, device=dev)
for name, fn, nbytes in (
        ("resize+normalize stretch", lambda: ops.preprocess(img, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], out_size=(S, S)), 3 * Hs * Ws + 12 * S * S),
        ("resize+normalize letterbox", lambda: ops.preprocess(img, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], out_size=(S, S), letterbox=True), 3 * Hs * Ws + 12 * S * S),
        ("resize u8 stretch", lambda: ops.resize_u8(img, (S, S)), 3 * Hs * Ws + 3 * S * S)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.profile_enable(64)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    k = L.profile_read()["vx_k_normalize/resize"]
    L.profile_disable()
    print("%-28s %.4f ms (min %.4f)  %.0f GB/s  = %.3f of 8 TB/s" % (name, k["avg_ms"], k["min_ms"], B * nbytes / k["avg_ms"] / 1e6,
                                                                   B * nbytes / k["avg_ms"] / 1e6 / 8000))
