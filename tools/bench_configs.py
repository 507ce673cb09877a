#!/usr/bin/env python3
"""The other BASELINE.json configurations as measured sub-records of bench.py's line (and as a CLI):
  fpn_ssd_640_b64             FPN-SSD 640x640, 20 classes, batch 64: normalize + prior-box decode + class-aware NMS
  retinaface_640_b128         RetinaFace 640x640, 5 landmarks, batch 128: normalize + prior-box decode + NMS
  nms_stress_100k_b32_top300  100 000 boxes/image, 80 classes, top-k 300, batch 32: NMS only
  yolo416_b256_dense          the headline pipeline at score_thr 0.06 (2 800 candidates/image)
  yolo416_b256_resize720p     the headline pipeline fed 720x1280 uint8 frames (resize fused into the normalize)
Each record carries images/s, per-kernel times (the library's HIP-event hooks), the roofline of its dominant
HBM-bound kernel, the NMS throughput in executed IoU tests/s against the VALU peak, and a time-boxed CPU baseline
(the oracle, on a bounded sample).  Inputs are synthetic (SURVEY §8d distributions) and resident in HBM."""
import collections
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# VALU peak for the IoU test: 256 CUs x 128 fp32 lanes/clk (the guide's 157.3 TFLOP/s vector peak / 2) x 2.4 GHz,
# divided by the 13 lane-operations of one test in the sweep's inner loop (4 min/max, 2 sub, 1 max, 2 mul, 2 add/sub,
# 1 sub, 1 max-accumulate; vx_async_cand in vx_nms.hip)
VALU_LANE_OPS_PER_S = 256 * 128 * 2.4e9
IOU_LANE_OPS = 13
IOU_PEAK_TESTS_PER_S = VALU_LANE_OPS_PER_S / IOU_LANE_OPS


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _spread(times_s, steps):
    """median / min / max of the repeats of one timed loop, in ms per step"""
    ms = sorted(t / steps * 1e3 for t in times_s)
    return {"repeats": len(ms), "median_ms_per_step": round(ms[len(ms) // 2], 4), "min_ms_per_step": round(ms[0], 4),
            "max_ms_per_step": round(ms[-1], 4)}


def timed_repeats(loop, steps, min_repeats=5, min_total_s=0.3, max_repeats=40):
    """loop(steps) = `steps` steps with every result materialised; repeated at least min_repeats times and until
    min_total_s of timed work has been seen.  Returns the per-repeat wall times (s)."""
    times = []
    while len(times) < min_repeats or (sum(times) < min_total_s and len(times) < max_repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(steps)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return times


def _timed(fn, steps, warmup, fn_async=None, ring=None, kernels=True, depth=2):
    """fn() = one synchronous step.  With fn_async (returns an object whose .result() materialises the step's
    detections) the timed loop keeps two batches in flight exactly as bench.py's headline loop does (--depth 2).
    Returns (last result, median seconds per step, kernel times, spread record)."""
    from vortex_amd import _lib as L
    last = [None]

    def loop(n):
        if fn_async is None:
            for _ in range(n):
                out = fn()
            last[0] = out
            return out
        pend = collections.deque()
        for i in range(n):
            if ring is not None:
                with ring.next():
                    cur = fn_async()
            else:
                cur = fn_async()
            pend.append(cur)
            if len(pend) >= depth:
                last[0] = pend.popleft().result()
        while pend:
            last[0] = pend.popleft().result()
        return last[0]
    torch.cuda.synchronize()
    loop(max(warmup, 1))
    times = timed_repeats(loop, steps)
    spread = _spread(times, steps)
    dt = spread["median_ms_per_step"] * 1e-3
    if not kernels:
        return last[0], dt, None, spread
    L.profile_enable(steps * 40)                      # separate pass: every kernel family bracketed by HIP events
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    kern = L.profile_read()
    L.profile_disable()
    return last[0], dt, {k: round(v["avg_ms"] * v["launches"] / steps, 5) for k, v in kern.items()}, spread


_RING = {}


def _multi_stream(fn, fn_async, steps, warmup, out, B, dt1):
    """The same loop with the batches on three HIP streams in turn (vortex_amd.streams.StreamRing) and four in flight:
    the pre-process and decode of the next batches run beside the one-workgroup-per-image NMS tail of this one.  Same
    results.  A/B on MI355X (ms per step, RetinaFace / FPN-SSD / stress top-300 / stress all; one stream 0.546 / 0.623 /
    0.134 / 0.940): 2 streams x 2 in flight 0.460 / 0.554 / 0.111 / 0.820; 2 x 3: 0.414 / 0.543 / 0.095 / 0.822;
    3 x 3: 0.429 / 0.520 / 0.104 / 0.794; 3 x 4: 0.404 / 0.512 / 0.097 / 0.790; 3 x 5, 3 x 6, 4 x 5: no better."""
    if _RING.get("off"):
        return None
    from vortex_amd.streams import StreamRing
    try:
        nring, depth = int(os.environ.get("VX_BENCH_RING", "3")), int(os.environ.get("VX_BENCH_RING_DEPTH", "4"))
        if "ring" not in _RING:                   # one set of streams per process, shared by every configuration
            _RING["ring"] = StreamRing(nring, _dev())
        out2, dt2, _, spread2 = _timed(fn, steps, warmup, fn_async, ring=_RING["ring"], kernels=False, depth=depth)
        if not (len(out) == len(out2) and all(torch.equal(a, b) for a, b in zip(out, out2))):
            return {"error": "two-stream results differ from the one-stream loop's"}
    except Exception as e:                        # the variant must not take the configuration's record down
        torch.cuda.synchronize()
        return {"error": "%s: %s" % (type(e).__name__, e)}
    return {"images_per_s": round(B / dt2, 1), "ms_per_step": round(dt2 * 1e3, 4), "spread": spread2,
            "hip_streams": len(_RING["ring"]), "batches_in_flight": depth,
            "speedup_vs_one_stream": round(dt1 / dt2, 3), "streams_overlap": bool(_RING["ring"].overlapping),
            "note": "batches go to the HIP streams of a vortex_amd.streams.StreamRing in turn; results checked equal to "
                    "the one-stream loop's; kernels_ms / roofline / nms of the record are from the one-stream pass"}


def _iou_tests(fn):
    """IoU tests the NMS sweeps execute on live lanes in ONE step (device counter, vx_profile_set_iou_counter)."""
    from vortex_amd import _lib as L
    cnt = torch.zeros(1, dtype=torch.int64, device=_dev())
    L.check(L.lib().vx_profile_set_iou_counter(cnt.data_ptr()), "vx_profile_set_iou_counter")
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        L.check(L.lib().vx_profile_set_iou_counter(None), "vx_profile_set_iou_counter")
    return int(cnt.item())


def _nms_record(tests, nms_ms):
    if not tests or not nms_ms:
        return None
    rate = tests / (nms_ms * 1e-3)
    return {"iou_tests_per_step": tests, "nms_kernels_ms": round(nms_ms, 5), "iou_tests_per_s": round(rate, 1),
            "valu_peak_tests_per_s": IOU_PEAK_TESTS_PER_S, "frac_of_valu_peak": round(rate / IOU_PEAK_TESTS_PER_S, 4),
            # a wave64 VALU instruction holds its SIMD for 4 cycles on gfx950 (tools/micro/sweep_loop.hip, SQ counters,
            # DESIGN 5.0b): the rate a kernel can reach is 64 lanes/clk/CU, half the figure above
            "frac_of_measured_valu_rate": round(rate / (IOU_PEAK_TESTS_PER_S / 2), 4),
            "note": "live-lane IoU tests counted on the device / summed NMS kernel time; peak = 256 CU x 128 lanes/clk x "
                    "2.4 GHz / 13 lane-ops per test"}


def _traffic(tag, kernel):
    """HBM bytes per launch of `kernel` in configuration `tag` from the newest committed profiles/traffic_r*.json
    (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `tools/bench_configs.py <tag>`, tools/prof_r02.sh)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "traffic_r*.json")), reverse=True):
        try:
            k = json.load(open(f)).get(tag, {}).get("kernels", {}).get(kernel)
        except Exception:
            k = None
        if k:
            return k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"], os.path.relpath(f, root)
    return None, None


def _roof(kernel, ms, algo_bytes, tag=None, pmc_kernel=None):
    ach = algo_bytes / (ms * 1e-3) / 1e9
    traffic, src = _traffic(tag, pmc_kernel or kernel) if tag else (None, None)
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "avg_kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": int(algo_bytes),
            "traffic": traffic,
            "traffic_source": ("%s: separate rocprofv3 --pmc passes of this configuration (FETCH_SIZE doubled as the guide "
                               "prescribes); not measured in this run" % src) if traffic else
                              "not collected (PMC passes are separate rocprofv3 runs: profiles/)"}


NMS_KERNELS = ("vx_k_nms_image", "vx_k_nms_build_keys", "vx_k_sort(keys1)+gather", "vx_k_nms_sweep", "vx_k_sort(keys2)+output",
               "vx_k_bm_pairs", "vx_k_bm_finish/scan")


def _cpu(fn, images, budget_s, what):
    """fn() processes `images` images on the host (the oracle); repeated while the budget lasts (at least once)."""
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * images / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s, %d run(s) x %d image(s), %.1f s" % (what, n, images, dt)}


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


def prior_cfg(tag, B, Lm, NL, sthr, steps=10, warmup=3, cpu_budget=4.0):
    import vortex_amd as V
    from vortex_amd import ops
    dev = _dev()
    P = 50400
    priors_h = make_priors(640)
    priors = priors_h.to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    NCOL = 4 + 2 * Lm + NL
    x = torch.randn(B, P, NCOL, generator=g, device=dev)
    x[..., 4 + 2 * Lm + 1:] = x[..., 4 + 2 * Lm + 1:] * 2 - 3
    img = torch.randint(0, 256, (B, 640, 640, 3), generator=g, dtype=torch.uint8, device=dev)
    norm = V.Normalize([0.5] * 3, [0.5] * 3).to(dev)
    pp = V.PriorBoxPostProcess(priors, n_landmarks=Lm).to(dev)
    st, it = torch.tensor([sthr]), torch.tensor([0.4])

    def fn():
        norm(img)
        return pp(x, st, it)

    def fn_async():
        norm(img)
        return pp.forward_async(x, st, it)
    cand = int(ops.prior_decode(x, priors, Lm, sthr).counts.sum())
    # two batches in flight like the headline loop.  forward_async takes the optimistic NMS call (vx_batched_nms_small):
    # the single-launch kernel, plus the class-partitioned path when 4..256 classes are declared (FPN-SSD); only a
    # single-class model whose images exceed 16384 candidates would lose that bet every time and switches it off
    pp.optimistic_nms = NL - 1 >= 4 or cand // B <= ops.NMS_SINGLE_LAUNCH_MAX
    out, dt, kern, spread = _timed(fn, steps, warmup, fn_async)
    multi = _multi_stream(fn, fn_async, steps, warmup, out, B, dt)
    # rows + priors once + the four output tensors (boxes 16, scores 4, labels 8, result rows 4 * (6 + 2L) bytes per
    # candidate; the staged records of rounds 1-2 are gone)
    algo = B * P * NCOL * 4 + P * 16 + cand * (52 + 8 * Lm)
    # which of the two decode kernels ran (vx_prior_decode picks the register-scored one when a row's logits sit in
    # its last 16-byte chunk)
    in_regs = NCOL % 4 == 0 and 8 <= NCOL <= 16 and 4 + 2 * Lm >= NCOL - 4
    rec = {"images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "spread": spread, "batch": B,
           "batches_in_flight": 2, "hip_streams": 1, "multi_stream": multi, "kernels_ms": kern,
           "candidates_per_image": cand // B, "detections_per_image": sum(int(r.shape[0]) for r in out) // B,
           "thresholds": {"score": sthr, "iou": 0.4},
           "roofline": _roof("vx_k_prior_decode_reg" if in_regs else "vx_k_prior_decode", kern["vx_k_prior_decode"], algo, tag)}
    tests = _iou_tests(fn)
    rec["nms"] = _nms_record(tests, sum(kern.get(k, 0.0) for k in NMS_KERNELS))
    if cpu_budget:
        from oracle import vortex_oracle as O
        xh = x[:1].cpu()
        imh = img[:1].cpu()
        rec["cpu_baseline"] = _cpu(lambda: (O.normalize(imh, [0.5] * 3, [0.5] * 3), O.prior_postprocess(xh, priors_h, st, it, Lm)),
                                   1, cpu_budget, "oracle normalize + prior-box decode + NMS on image 0 of the same batch")
    return rec


def stress_cfg(steps=10, warmup=3, cpu_budget=8.0, top_k=300, tag="nms_stress_100k_b32_top300"):
    """BASELINE configs[4]: 100 000 boxes per image, 80 classes, score_thr 0.001, top-k 300 (None: every kept box),
    batch 32, NMS only.  The score threshold is applied the way a decoder applies it -- candidates that pass come
    first, `counts` says how many -- once, outside the timed loop (the boxes are this configuration's INPUT); the timed
    step is the NMS call + packing + the read-back of counts and the per-image result tuple, two batches in flight."""
    from vortex_amd import ops
    from vortex_amd.postprocess import PendingDetections
    dev = _dev()
    B, K, C, sthr, ithr = 32, 100_000, 80, 0.001, 0.45
    g = torch.Generator(device=dev).manual_seed(9)
    c = torch.rand(B, K, 2, generator=g, device=dev)
    wh = torch.rand(B, K, 2, generator=g, device=dev) * 0.28 + 0.02
    boxes = torch.cat((c - wh / 2, c + wh / 2), 2).contiguous()
    scores = torch.rand(B, K, generator=g, device=dev)
    cls = torch.randint(0, C, (B, K), generator=g, device=dev, dtype=torch.int64)
    passed = scores > sthr
    order = torch.sort((~passed).to(torch.int8), dim=1, stable=True).indices          # passing candidates first, row order kept
    boxes = torch.gather(boxes, 1, order[..., None].expand(-1, -1, 4)).contiguous()
    scores, cls = torch.gather(scores, 1, order).contiguous(), torch.gather(cls, 1, order).contiguous()
    counts = passed.sum(1).to(torch.int32)
    det = torch.cat((boxes, scores[..., None], cls[..., None].float()), 2).contiguous()

    def fn_async():
        meta = torch.zeros(2 * B + 1, dtype=torch.int32, device=dev)
        meta[:B] = counts
        # optimistic: no class holds more than 16384 boxes; with a top_k the score-prefix path does the work
        _, out_det, keep_counts, _ = ops.batched_nms(boxes, scores, cls, det, meta[:B], ithr, C, top_k, meta=meta, optimistic=True)
        return PendingDetections.packing(out_det, keep_counts, meta)

    def fn():
        return fn_async().result()
    out, dt, kern, spread = _timed(fn, steps, warmup, fn_async)
    multi = _multi_stream(fn, fn_async, steps, warmup, out, B, dt)
    rec = {"images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "spread": spread, "batch": B,
           "batches_in_flight": 2, "hip_streams": 1, "multi_stream": multi, "kernels_ms": kern,
           "boxes_per_image": K, "candidates_per_image": int(counts.sum()) // B, "classes": C, "top_k": top_k,
           "thresholds": {"score": sthr, "iou": ithr},
           "detections_per_image": sum(int(r.shape[0]) for r in out) // B}
    rec["nms"] = _nms_record(_iou_tests(fn), sum(kern.get(k, 0.0) for k in NMS_KERNELS))
    # NMS is VALU-bound, not HBM-bound: its roofline is the IoU-test rate above; the bytes it must move
    # (keys, boxes, output rows) are reported for completeness
    if rec["nms"]:
        rec["roofline"] = {"bound": "valu", "kernel": "NMS kernels (see nms.nms_kernels_ms)", "achieved": rec["nms"]["iou_tests_per_s"],
                           "peak": IOU_PEAK_TESTS_PER_S, "unit": "IoU tests/s", "frac": rec["nms"]["frac_of_valu_peak"]}
    if top_k:
        rec["note"] = ("top_k on a large input takes the score-prefix path: the kept boxes among a score prefix of the "
                       "candidates are the head of the full keep list, so ~2 500 of the 100 000 boxes per image decide the "
                       "result (vx_k_prefix_thr / vx_k_prefix_compact); the configuration without top_k below measures the "
                       "full pair work")
    if cpu_budget:
        from oracle import vortex_oracle as O
        n0 = int(counts[0])
        bh, sh, ch = boxes[0, :n0].cpu(), scores[0, :n0].cpu(), cls[0, :n0].cpu()
        rec["cpu_baseline"] = _cpu(lambda: O.batched_nms(bh, sh, ch, torch.tensor([ithr]))[:top_k], 1, cpu_budget,
                                   "oracle batched_nms (single-thread C greedy NMS) on image 0 of the same batch, every "
                                   "candidate above the score threshold%s" % (", then the first %d" % top_k if top_k else ""))
    return rec


def latency_cfg(steps=200, warmup=20, cpu_budget=4.0):
    """BASELINE configs[0] shape: YOLOv3 416x416, 80 classes, batch 1 -- the reference's own CPU-runnable case.  Reports
    the synchronous latency of one image (pre-process + fused decode + NMS + result tuple) eagerly (two module calls),
    through the one-call step (vx_yolo_pipeline_step, buffers reused) and as a HIP-graph replay, beside the oracle's time for the same image on the host."""
    import bench
    import vortex_amd as V
    from vortex_amd.graphs import GraphedPostProcess
    dev = _dev()
    S = 416
    img, heads = bench.synth_on_device(1, dev, seed=1234)
    norm = V.Normalize().to(dev)
    post = V.YoloV3FusedPostProcess(img_size=S)
    st, it = torch.tensor([0.9]), torch.tensor([0.2])

    def eager(n):
        for _ in range(n):
            norm(img)
            out = post(heads, st, it)
        return out
    graphed = GraphedPostProcess(post, heads, st, it, preprocess=norm, images=img)
    one_call = V.YoloV3PipelineStep(norm, post)

    def step(n):
        for _ in range(n):
            out = one_call(img, heads, st, it)[1]
        return out

    def graph(n):
        for _ in range(n):
            out = graphed()
        return out
    rec = {"batch": 1, "thresholds": {"score": 0.9, "iou": 0.2}, "mode": "synchronous: every step returns its result tuple"}
    for name, loop in (("eager", eager), ("one_call", step), ("hipgraph", graph)):
        loop(warmup)
        sp = _spread(timed_repeats(loop, steps), steps)
        rec[name] = {"latency_ms": sp["median_ms_per_step"], "images_per_s": round(1e3 / sp["median_ms_per_step"], 1), "spread": sp}
    # the record's own figure: the one-call step (no graph capture needed); the two-module-call figure stays beside it
    rec["images_per_s"] = rec["one_call"]["images_per_s"]
    rec["ms_per_step"] = rec["one_call"]["latency_ms"]
    rec["detections_per_image"] = int(eager(1)[0].shape[0])
    del graphed
    if cpu_budget:
        from oracle import vortex_oracle as O
        hh = [h.cpu() for h in heads]
        ih = img.cpu()
        rec["cpu_baseline"] = _cpu(lambda: (O.normalize(ih, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
                                            O.yolo_postprocess(O.yolo_head_eval(hh, S), st, it, S)), 1, cpu_budget,
                                   "oracle pre-process + head decode + decoder + NMS on the same image")
    return rec


def yolo_cfg(kind, steps=20, warmup=5, cpu_budget=3.0, variants=True):
    """kind = 'dense' (score_thr 0.06) or 'resize720p' (720x1280 frames resized inside the pre-process)."""
    import bench
    import vortex_amd as V
    from vortex_amd import ops
    dev = _dev()
    B, S, C = 256, 416, 80
    img, heads = bench.synth_on_device(B, dev, seed=4321)
    sthr = 0.06 if kind == "dense" else 0.9
    st, it = torch.tensor([sthr]), torch.tensor([0.2])
    post = V.YoloV3FusedPostProcess(img_size=S)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    if kind == "resize720p":
        g = torch.Generator(device=dev).manual_seed(11)
        frames = torch.randint(0, 256, (B, 720, 1280, 3), generator=g, dtype=torch.uint8, device=dev)

        def pre():
            return ops.preprocess(frames, mean, std, 255, out_size=(S, S))
    else:
        norm = V.Normalize(mean, std).to(dev)

        def pre():
            return norm(img)

    def fn():
        pre()
        return post(heads, st, it)

    def fn_async():
        pre()
        return post.forward_async(heads, st, it)
    out, dt, kern, spread = _timed(fn, steps, warmup, fn_async)
    multi = _multi_stream(fn, fn_async, steps, warmup, out, B, dt)
    cand = int(ops.yolo_fused_decode(heads, S, sthr).counts.sum())
    tag = "yolo416_b256_" + kind
    rec = {"images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "spread": spread, "batch": B,
           "batches_in_flight": 2, "hip_streams": 1, "multi_stream": multi, "kernels_ms": kern,
           "candidates_per_image": cand // B, "detections_per_image": sum(int(r.shape[0]) for r in out) // B,
           "thresholds": {"score": sthr, "iou": 0.2}}
    if kind == "resize720p":
        rec["roofline"] = _roof("vx_k_resize_fast<true> (reported as vx_k_normalize/resize)", kern["vx_k_normalize/resize"],
                                B * (3 * 720 * 1280 + 12 * S * S), tag, "vx_k_resize_fast<true>")
    else:
        rec["roofline"] = _roof("vx_k_yolo_fused", kern["vx_k_yolo_fused"], B * bench.ALGO_BYTES_DECODE + 32 * cand, tag)
    rec["nms"] = _nms_record(_iou_tests(fn), sum(kern.get(k, 0.0) for k in NMS_KERNELS))
    if kind == "resize720p" and variants:
        # the letterbox mode of the same kernel (aspect kept: 416 x 234 of the 416 x 416 output is resized, the rest is
        # the pad value) and the plain u8 -> u8 resize of runtime.resize_batch, kernel time only
        def _kernel_ms(f):
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20
        sh = min(S / 720.0, S / 1280.0)
        Hl, Wl = int(round(720 * sh)), int(round(1280 * sh))
        lb_ms = _kernel_ms(lambda: ops.preprocess(frames, mean, std, 255, out_size=(S, S), letterbox=True))
        lb_bytes = B * (3 * 720 * 1280 + 12 * S * S)
        u8_out = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev)
        u8_ms = _kernel_ms(lambda: ops.resize_u8(frames, (S, S), out=u8_out))
        u8_bytes = B * (3 * 720 * 1280 + 3 * S * S)
        rec["variants"] = {
            "letterbox": {"kernel_ms": round(lb_ms, 5), "resized_area": [Hl, Wl], "algorithmic_bytes_per_launch": lb_bytes,
                          "frac_of_8TBps": round(lb_bytes / (lb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "resize_u8_only": {"kernel_ms": round(u8_ms, 5), "algorithmic_bytes_per_launch": u8_bytes,
                               "frac_of_8TBps": round(u8_bytes / (u8_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    if cpu_budget:
        from oracle import vortex_oracle as O
        n = 2
        hh = [h[:n].cpu() for h in heads]
        if kind == "resize720p":
            fr = frames[:n].cpu().numpy()

            def cpu():
                import numpy as np
                x = torch.from_numpy(np.stack([O.resize_bilinear_u8(f, S, S) for f in fr]))
                return O.normalize(x, mean, std), O.yolo_postprocess(O.yolo_head_eval(hh, S), st, it, S)
        else:
            ih = img[:n].cpu()

            def cpu():
                return O.normalize(ih, mean, std), O.yolo_postprocess(O.yolo_head_eval(hh, S), st, it, S)
        rec["cpu_baseline"] = _cpu(cpu, n, cpu_budget, "oracle pre-process + head decode + decoder + NMS on images 0-1 of the same batch")
    return rec


CONFIGS = {
    "fpn_ssd_640_b64": lambda **kw: prior_cfg("fpn_ssd_640_b64", 64, 0, 21, 0.3, **kw),
    "retinaface_640_b128": lambda **kw: prior_cfg("retinaface_640_b128", 128, 5, 2, 0.5, **kw),
    "nms_stress_100k_b32_top300": lambda **kw: stress_cfg(**kw),
    "nms_stress_100k_b32_all": lambda **kw: stress_cfg(top_k=None, tag="nms_stress_100k_b32_all", **kw),
    "yolo416_b1_latency": lambda **kw: latency_cfg(**kw),
    "yolo416_b256_dense": lambda **kw: yolo_cfg("dense", **kw),
    "yolo416_b256_resize720p": lambda **kw: yolo_cfg("resize720p", **kw),
    "yolo416_b256_resize720p_only": lambda **kw: yolo_cfg("resize720p", variants=False, **kw),   # profiling: the stretch kernel alone
}


def run_configs(names=None, cpu=True):
    out = {}
    for name in (names or [n for n in CONFIGS if not n.endswith("_only")]):
        kw = {} if cpu else {"cpu_budget": 0}
        try:
            out[name] = CONFIGS[name](**kw)
        except Exception as e:                                  # a failed sub-record must not take the headline line down
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    alias = {"ssd": "fpn_ssd_640_b64", "retinaface": "retinaface_640_b128", "stress": "nms_stress_100k_b32_top300",
             "dense": "yolo416_b256_dense", "resize": "yolo416_b256_resize720p", "stress_all": "nms_stress_100k_b32_all",
             "b1": "yolo416_b1_latency", "resize_only": "yolo416_b256_resize720p_only"}
    _RING["off"] = "--one-stream" in sys.argv      # profiling runs: keep overlapped launches out of the kernel statistics
    args = [a for a in sys.argv[1:] if a not in ("--no-cpu", "--one-stream")]
    which = [alias.get(a, a) for a in args] or None
    for k, v in run_configs(which, cpu="--no-cpu" not in sys.argv).items():
        print(json.dumps({"config": k, **v}), flush=True)
