#!/usr/bin/env python3
"""Phase breakdown of vx_k_nms_image on the prior-box configs (RetinaFace b128 / FPN-SSD b64 candidates)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench_configs import make_priors  # noqa: E402
from vortex_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "retinaface"
if which == "dense":                                  # the headline pipeline's candidates at score_thr 0.06
    import bench
    B, NL, ithr = 256, 81, 0.2
    _, heads = bench.synth_on_device(B, dev, seed=4321)
    c = ops.yolo_fused_decode(heads, 416, 0.06)
else:
    B, Lm, NL, sthr = (128, 5, 2, 0.5) if which == "retinaface" else (64, 0, 21, 0.3)
    ithr = 0.4
    P = 50400
    priors = make_priors(640).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, P, 4 + 2 * Lm + NL, generator=g, device=dev)
    x[..., 4 + 2 * Lm + 1:] = x[..., 4 + 2 * Lm + 1:] * 2 - 3
    c = ops.prior_decode(x, priors, Lm, sthr)
lib = L.lib()
K = c.R
D = c.D
nb = lib.vx_batched_nms_workspace_bytes(B, K)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
keep_idx = torch.empty((B, K), dtype=torch.int64, device=dev)
out_det = torch.empty((B, K, D), device=dev)
L.profile_enable(64)
for it in range(3):
    rc = lib.vx_batched_nms(c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(), c.det.data_ptr(),
                            c.counts.data_ptr(), B, K, D, NL - 1, ithr, 0, keep_idx.data_ptr(), out_det.data_ptr(),
                            c.keep_counts.data_ptr(), K, c.status.data_ptr(), ws.data_ptr(), nb, L.stream_ptr(dev))
    assert rc == 0
torch.cuda.synchronize()
print({k: round(v["avg_ms"], 4) for k, v in L.profile_read().items()})
boff = lib.vx_batched_nms_bm_counters_offset(B, K)
if boff:
    cw = ws[boff:boff + 256].view(torch.int32).cpu().tolist()
    print("bitmask path: segments %d, work items %d; debug [items, tile pairs, visits, trips, hit boxes, edges] = %s"
          % (cw[0], cw[1], cw[8:14]))
    st5 = cw[16:21]
    print("runs_bucket stamps (ticks between phases: tables, gather, merge rounds, rows):", [st5[i + 1] - st5[i] for i in range(4)])
off = lib.vx_batched_nms_stamps_offset(B, K)
SW = 136
allw = ws[off:off + B * SW * 8].view(torch.int64).view(B, SW).cpu().double()
st = allw[:, :8]
wv = allw[:, 8:].view(B, 16, 8)
print("per-wave cycles of the column-owner sweep, image 0: [wait, apply, chain, rows, total, tests]")
for w_ in range(16):
    print("  wave %2d " % w_, [int(v) for v in wv[0, w_, :8]])
print("mean over images and waves:", [int(v) for v in wv[:, :, :8].mean(dim=(0, 1))])
ss = allw[:, 16:19]
print("vx_bm_scan phases (mean ticks): edge sort %.0f  chain %.0f;  edges/image %s" % ((ss[:, 1] - ss[:, 0]).mean(), (ss[:, 2] - ss[:, 1]).mean(), ""))
print("vx_bm_chain sections (mean ticks): dead-mask wait %.0f  resolve %.0f  count/kept %.0f  cross edges + turn %.0f" % tuple(allw[:, 19:23].mean(0).tolist()))
ps = allw[:, 24:32]
print("vx_bm_prep phases (mean ticks): zero %.0f  extent+copy %.0f  x histogram %.0f  scan %.0f  slab pass %.0f  scan2 %.0f  scatter %.0f  tile boxes %.0f" % tuple([0.0] + [(ps[:, i + 1] - ps[:, i]).mean() for i in range(7)]))
fs = allw[:, 8:12]
print("vx_k_bm_finish phases (mean ticks): scan %.0f  emit %.0f  tail %.0f" % tuple((fs[:, i + 1] - fs[:, i]).mean() for i in range(3)))
d = (st[:, 1:] - st[:, :-1])
names = ["keys", "sort1", "gather", "sweep-wave", "sweep-block", "load+sort2", "output"]
print("candidates/img %.0f (max %d) kept/img %.0f" % (c.counts.float().mean(), int(c.counts.max()), c.keep_counts.float().mean()))
tot = (st[:, 7] - st[:, 0])
print("total ticks: mean %.0f max %.0f" % (tot.mean(), tot.max()))
for i, n in enumerate(names):
    print("%-12s mean %8.0f  max %8.0f  share %.1f%%" % (n, d[:, i].mean(), d[:, i].max(), 100 * d[:, i].mean() / tot.mean()))
