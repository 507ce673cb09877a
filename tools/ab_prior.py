#!/usr/bin/env python3
"""K4 alone (vx_prior_decode + cand_finish) on the two prior-box configs: time, GB/s, at the config's threshold and
with a threshold nobody passes (streaming ceiling of the kernel's structure)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench_configs import make_priors  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
P = 50400
priors = make_priors(640).to(dev)
for name, B, Lm, NL, sthr in (("retinaface", 128, 5, 2, 0.5), ("fpn_ssd", 64, 0, 21, 0.3)):
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, P, 4 + 2 * Lm + NL, generator=g, device=dev)
    x[..., 4 + 2 * Lm + 1:] = x[..., 4 + 2 * Lm + 1:] * 2 - 3
    for thr in (sthr, 2.0):
        for _ in range(3):
            c = ops.prior_decode(x, priors, Lm, thr)
        torch.cuda.synchronize()
        L.profile_enable(64)
        for _ in range(20):
            c = ops.prior_decode(x, priors, Lm, thr)
        torch.cuda.synchronize()
        k = L.profile_read()
        L.profile_disable()
        ms = k["vx_k_prior_decode"]["avg_ms"]
        cand = int(c.counts.sum())
        nbytes = x.numel() * 4 + P * 16 + cand * (32 + 8 * Lm)
        print("%-10s thr %.1f: prior_decode %.4f ms (min %.4f)  %.0f GB/s  cand/img %d  finish %.4f ms" % (
            name, thr, ms, k["vx_k_prior_decode"]["min_ms"], nbytes / ms / 1e6, cand // B, k["vx_k_cand_finish"]["avg_ms"]))
