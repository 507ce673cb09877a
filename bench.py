#!/usr/bin/env python3
"""bench.py — images/s of the detection pre-process + decode + NMS path on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
    K1  uint8 NHWC images  -> fp32 NCHW (normalise)                      vx_preprocess_u8hwc_f32nchw
    K23 three raw YOLOv3 head tensors -> thresholded candidates          vx_yolo_fused_decode
    NMS class-aware batched NMS -> per-image detection lists             vx_batched_nms + vx_pack_rows
    + the API's single D2H read of the per-image counts and the tuple of per-image tensors.
Workload: BASELINE.json configs[3] — YOLOv3 416x416, 80 classes, batch 256 per GPU (weak scaling:
every rank processes its own 256-image shard; no data-path collective unless --allgather).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N>1 the driver launches it under torch.distributed.run (one rank per GPU, RCCL); started WITHOUT a launcher
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) it spawns the N ranks itself -- fresh child
processes, before anything touches a GPU -- over RCCL when the node has N GPUs, else over gloo with ranks
sharing the GPUs there are (rehearsal; the line says `"backend": "gloo"`).

At N=1 the line also carries `configs`: the other BASELINE.json configurations (FPN-SSD b64, RetinaFace b128,
NMS stress 100k, dense YOLO, 720p resize), each with images/s, roofline, NMS IoU-test throughput and a time-boxed
CPU baseline (tools/bench_configs.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIG_NAMES = ("fpn_ssd_640_b64", "retinaface_640_b128", "nms_stress_100k_b32_top300", "nms_stress_100k_b32_all",
                "yolo416_b256_dense", "yolo416_b256_resize720p", "yolo416_b1_latency")
S, C = 416, 80
R = 10647
ROW_BYTES = (5 + C) * 4
ALGO_BYTES_PRE = 3 * S * S + 12 * S * S                 # 2 595 840 B / image   
ALGO_BYTES_DECODE = R * ROW_BYTES                       # 3 619 980 B / image   (+52 B per survivor)
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synth_on_device(B, dev, seed):
    """Same distributions as oracle.synth_yolo_heads / synth_images, generated on the GPU."""
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8, device=dev)
    heads = []
    for G in (S // 32, S // 16, S // 8):
        t = torch.randn(B, 3, 5 + C, G, G, generator=g, device=dev)
        t[:, :, 2:4] *= 0.5
        t[:, :, 4] = t[:, :, 4] * 2 - 4
        t[:, :, 5:] = t[:, :, 5:] * 1.5 - 2
        heads.append(t.view(B, 3 * (5 + C), G, G).contiguous())
    return img, heads


def cpu_baseline(sthr, ithr, budget_s=15.0, n_img=8):
    """The oracle (CPU port of the reference path: torch-CPU ops one for one, per-image Python loop,
    single-thread C greedy NMS) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import vortex_oracle as O
    img = O.synth_images(n_img, S, seed=0)
    heads = O.synth_yolo_heads(n_img, S, C, seed=0)
    st, it = torch.tensor([sthr]), torch.tensor([ithr])

    def run():
        x = O.normalize(img, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        pred = O.yolo_head_eval(heads, S)
        return x, O.yolo_postprocess(pred, st, it, S)

    def timed(budget):
        run()
        t0 = time.perf_counter()
        n = 0
        while True:
            run()
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        return n, time.perf_counter() - t0

    n, dt = timed(budget_s)                                  # all the threads torch uses on this host
    cores = torch.get_num_threads()
    torch.set_num_threads(1)                                 # also report the single-thread rate
    n1, dt1 = timed(budget_s / 3)
    torch.set_num_threads(cores)
    return {"value": n * n_img / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "value_1_thread": round(n1 * n_img / dt1, 2),
            "sample": "%d iterations x %d images of the same YOLOv3-416/80-class synthetic workload, "
                      "normalize + head decode + decoder + NMS, score_thr=%g iou_thr=%g (%.0f s; then %d "
                      "iterations on 1 thread, %.0f s)" % (n, n_img, sthr, ithr, dt, n1, dt1)}


def count_gpus_without_hip():
    """GPUs visible to the ranks, counted WITHOUT touching torch.cuda / HIP in this launcher process (on ROCm
    torch.cuda.device_count() can fall through to hipGetDeviceCount, which initialises the runtime in the parent the
    rank processes are then started from): the KFD topology lists one node per agent, GPUs are those with SIMDs; the
    visibility variables narrow that."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                pass
    except OSError:
        return 0
    return n


def spawn_ranks(n, argv):
    """Self-contained multi-GPU launch: N fresh worker processes (this process never initialises a GPU), one rank
    per GPU, rendezvous on 127.0.0.1.  Fewer GPUs than ranks -> gloo backend, ranks share GPUs (rehearsal only)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ndev = count_gpus_without_hip()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if ndev < n:
            env.setdefault("VX_BENCH_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    return rc


def load_traffic(tag):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (separate runs,
    profiles/traffic_rNN.json, the newest round; corrected as MI355X_MICROARCH.md §HBM prescribes) and the file it
    came from: (None, None) if not measured.  The number is NOT measured in this run; the line says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return d.get(tag, {}).get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--score-thr", type=float, default=0.9)    # experiments/configs/yolov3_darknet_608.yml:57-58
    ap.add_argument("--iou-thr", type=float, default=0.2)
    ap.add_argument("--dense", action="store_true", help="score_thr 0.06 (tests/config/test_detection.yml:85-86)")
    ap.add_argument("--unfused", action="store_true", help="API-boundary pipeline: YoloV3Layer x3 -> YoloV3PostProcess")
    ap.add_argument("--allgather", action="store_true", help="RCCL all-gather of the detection lists every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra pass that times the opt-in variants")
    ap.add_argument("--configs", default="all",
                    help="N=1 only: the other BASELINE configs measured as sub-records of the line: 'all', 'none' or a "
                         "comma-separated list of " + ", ".join(CONFIG_NAMES))
    ap.add_argument("--repeats", type=int, default=5, help="repeats of the timed region (median reported)")
    ap.add_argument("--rotate-images", type=int, default=1,
                    help="number of distinct uint8 image batches fed in turn (1: the same batch every step, which then "
                         "stays in the 256 MB Infinity Cache between steps; 4+: every step reads its images from HBM)")
    ap.add_argument("--depth", type=int, default=2,
                    help="batches in flight: 1 = synchronous API call per step (forward()), 2 = forward_async(): "
                         "the launches of step i+1 are enqueued before the results of step i are collected")
    ap.add_argument("--streams", type=int, default=1,
                    help="2 = the pre-process runs on its own HIP stream, ordered after the decode kernel of the same "
                         "step, so the latency-bound NMS/pack tail overlaps the HBM-bound normalize; 1 = one stream")
    ap.add_argument("--head-launches", action="store_true",
                    help="--unfused only: the three YoloV3Layer decodes as three launches (vx_yolo_head_decode) instead "
                         "of one (vx_yolo_head_decode3)")
    ap.add_argument("--batch-streams", type=int, default=1,
                    help="N > 1: whole steps go to N HIP streams in turn (needs --depth 2), so the kernels of step i+1 "
                         "run beside the tail of step i wherever that tail leaves CUs or HBM bandwidth idle")
    ap.add_argument("--threshold-first", action="store_true",
                    help="opt-in decode variant (nan_check='survivors'): objectness plane first, other planes only "
                         "for surviving rows; same detections, NaN check restricted -> not the reference-exact line")
    args = ap.parse_args()
    if args.dense:
        args.score_thr = 0.06

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, as child processes, before any GPU call in this process
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus (or without a launcher)"
    dist = None
    # rehearsal knob (tests on a one-GPU box): VX_BENCH_BACKEND=gloo lets several ranks share a GPU
    backend = os.environ.get("VX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import vortex_amd as V
    from vortex_amd import _lib as L, ops, parallel

    B = args.batch
    img, heads = synth_on_device(B, dev, seed=1234 + rank)
    imgs = [img] + [synth_on_device(B, dev, seed=4321 + 17 * k + rank)[0] for k in range(1, max(1, args.rotate_images))]
    turn = [0]
    norm = V.Normalize().to(dev)
    sthr, ithr = torch.tensor([args.score_thr]), torch.tensor([args.iou_thr])
    if args.unfused:
        layers = []
        off = 0
        for h, mask in zip(heads, ((6, 7, 8), (3, 4, 5), (0, 1, 2))):
            G = h.shape[-1]
            layers.append((V.YoloV3Layer(S, (G, G), list(mask), ops.COCO_ANCHORS, C).to(dev).eval(), off))
            off += 3 * G * G
        head3 = V.YoloV3HeadDecode(S, C).to(dev).eval() if not args.head_launches else None
        post = V.YoloV3PostProcess(img_size=S)
        decoded = torch.empty((B, R, 5 + C), device=dev)
    else:
        post = V.YoloV3FusedPostProcess(img_size=S, nan_check="survivors" if args.threshold_first else "full")

    two_streams = args.streams > 1 and not args.unfused and not args.allgather
    if two_streams:
        # NMS + packing of a batch on a second stream behind the decode (vx_yolo_fused_postprocess_overlapped): the
        # current stream carries only K1 and the fused decode, back to back
        from vortex_amd.streams import overlapping_stream
        post.tail_stream = overlapping_stream(torch.cuda.current_stream(dev))    # HIP maps streams to hardware queues unevenly

    def step():
        turn[0] = (turn[0] + 1) % len(imgs)
        x = norm(imgs[turn[0]])                             # K1 (consumed by the backbone, out of scope)
        inp = heads
        if args.unfused:
            if head3 is not None:
                head3(*heads, out=decoded)                  # YoloV3Layer x3 + concat, one launch
            else:
                for (layer, off), h in zip(layers, heads):
                    layer(h, out=decoded, row_offset=off)
            inp = decoded
        if args.allgather:
            packed, meta = post.forward_device(inp, sthr, ithr)
            t_ag = time.perf_counter()
            res = parallel.allgather_detections(packed, meta, B)     # phase 1 waits for this step's GPU work (count D2H)
            ag_s[0] += time.perf_counter() - t_ag
            return x, res
        pend = post.forward_async(inp, sthr, ithr)          # counts land in pinned host memory, not awaited
        return x, pend if args.depth > 1 else pend.result()

    ag_s = [0.0]            # host time inside allgather_detections (includes the wait for the step's kernels)

    ring = V.StreamRing(args.batch_streams, dev) if args.batch_streams > 1 else None

    def run_steps(n):
        """n full steps; every step's per-image tuple is materialised before this returns."""
        out, prev = None, None
        for _ in range(n):
            if ring is not None and args.depth > 1 and not args.allgather:
                with ring.next():
                    cur = step()
            else:
                cur = step()
            if args.depth > 1 and not args.allgather:
                if prev is not None:
                    out = (prev[0], prev[1].result())
                prev = cur
            else:
                out = cur
        if prev is not None:
            out = (prev[0], prev[1].result())
        return out

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = run_steps(max(args.warmup, 1))
    n_det = sum(int(r.shape[0]) for r in out[1]) if not args.allgather else None

    # Inside the timed region only the dominant kernel is bracketed by HIP events (each event pair costs a few
    # microseconds of stream time: bracketing all five kernels costs ~12 % of the step); the other kernels
    # are timed in a separate untimed pass below.
    profile = not args.no_profile
    dom = "vx_k_yolo_decode_rows" if args.unfused else "vx_k_yolo_fused"
    if profile:
        L.profile_enable(max(args.steps, 16) * 42, only=[dom])   # one launch of the dominant kernel per step, <= 40 repeats
    # The timed region (exactly --steps steps between two barriers) is repeated: `value` is the MEDIAN repeat, the line
    # carries min / max (`spread`).  One rank: at least 5 repeats and 0.3 s of timed work; several ranks: 5 repeats each,
    # every repeat bracketed by the barrier, the slowest rank's time counting for it.
    rep_dt = []
    ag_s[0] = 0.0
    while len(rep_dt) < args.repeats or (world == 1 and sum(rep_dt) < 0.3 and len(rep_dt) < 40):
        barrier()
        t0 = time.perf_counter()
        out = run_steps(args.steps)
        barrier()
        rep_dt.append(time.perf_counter() - t0)
    ag_ms_per_step = ag_s[0] / (len(rep_dt) * args.steps) * 1e3
    kern = L.profile_read() if profile else {}
    if profile:
        L.profile_enable(max(args.steps * 4, 16))          # untimed pass: every kernel family
        run_steps(args.steps)
        torch.cuda.synchronize()
        kern_all = L.profile_read()
        kern_all[dom] = kern.get(dom, kern_all.get(dom))
        L.profile_disable()
    else:
        kern_all = {}

    # Opt-in decode variant, reported beside the contract line (never as `value`): same detections, but the NaN
    # check of the reference is restricted to objectness + surviving rows, so most of the head bytes are not read.
    variants = {}
    if world == 1 and not (args.unfused or args.allgather or args.threshold_first or args.no_variants):
        post_full = post
        post = V.YoloV3FusedPostProcess(img_size=S, nan_check="survivors")
        run_steps(3)
        L.profile_enable(max(args.steps * 4, 16), only=[dom])
        barrier()
        t1 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        dtv = time.perf_counter() - t1
        kv = L.profile_read().get(dom, {})
        L.profile_disable()
        variants["threshold_first"] = {
            "value": round(B * args.steps / dtv, 1), "unit": "images/s", "ms_per_step": round(dtv / args.steps * 1e3, 4),
            "decode_kernel_ms": round(kv.get("avg_ms", 0.0), 5),
            "note": "YoloV3FusedPostProcess(nan_check='survivors'): identical detections; NaN check covers objectness "
                    "and surviving rows only (deviation from postprocess/yolov3.py:35-36), hence not the headline"}
        # the same variant replayed as one HIP graph (vortex_amd/graphs.py): at 0.15 ms of GPU work per step the
        # ~0.09 ms of launch + allocation calls per eager step is what limits the line above
        from vortex_amd.graphs import GraphedPostProcess
        graphed = GraphedPostProcess(post, heads, sthr, ithr, preprocess=norm, images=img)

        def graph_steps(n):
            prev = None
            for _ in range(n):
                cur = graphed.launch()
                if prev is not None:
                    prev.result()
                prev = cur
            return prev.result()

        graph_steps(3)
        barrier()
        t1 = time.perf_counter()
        graph_steps(args.steps)
        barrier()
        dtg = time.perf_counter() - t1
        variants["threshold_first_hipgraph"] = {
            "value": round(B * args.steps / dtg, 1), "unit": "images/s", "ms_per_step": round(dtg / args.steps * 1e3, 4),
            "note": "same kernels, pre-process + post-process captured once and replayed (fixed shapes and thresholds)"}
        del graphed
        post = post_full
        # the headline pipeline fed from four image batches in turn: with ONE batch the 133 MB of uint8 images stay in the
        # 256 MB Infinity Cache between steps and K1 reads them from there (K1 0.098 ms); four batches come from HBM
        if len(imgs) == 1:
            imgs.extend(synth_on_device(B, dev, seed=4321 + 17 * k + rank)[0] for k in range(1, 4))
            run_steps(3)
            L.profile_enable(max(args.steps * 4, 16), only=["vx_k_normalize/resize"])
            barrier()
            t1 = time.perf_counter()
            run_steps(args.steps)
            barrier()
            dtr = time.perf_counter() - t1
            kr = L.profile_read().get("vx_k_normalize/resize", {})
            L.profile_disable()
            variants["rotating_image_batches"] = {
                "value": round(B * args.steps / dtr, 1), "unit": "images/s", "ms_per_step": round(dtr / args.steps * 1e3, 4),
                "preprocess_kernel_ms": round(kr.get("avg_ms", 0.0), 5),
                "note": "the headline step with 4 distinct uint8 image batches fed in turn (every step reads its images "
                        "from HBM, not from the Infinity Cache)"}
            del imgs[1:]

    rank_devices = [local]
    if dist is not None:
        td = torch.tensor([local], device=dev if backend == "nccl" else "cpu", dtype=torch.int64)
        alld = [torch.zeros_like(td) for _ in range(world)]
        dist.all_gather(alld, td)
        rank_devices = [int(v.item()) for v in alld]
    per_rank = [rep_dt]
    if dist is not None:
        t = torch.tensor(rep_dt, device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [[float(v) for v in a_.tolist()] for a_ in allt]
    job_dt = sorted(max(r[i] for r in per_rank) for i in range(len(rep_dt)))     # whole-job time of a repeat = its slowest rank
    dt = job_dt[len(job_dt) // 2]
    rank_dt = [sorted(r)[len(r) // 2] for r in per_rank]

    result = None
    if rank == 0:
        n_gpus = world
        ips = n_gpus * B * args.steps / dt
        meta = out[1] if args.allgather else None
        cand = int(ops.yolo_fused_decode(heads, S, args.score_thr).counts.sum().item())
        roof = None
        if dom in kern:
            avg_ms = kern[dom]["avg_ms"]
            algo = B * ALGO_BYTES_DECODE + 32 * cand
            ach = algo / (avg_ms * 1e-3) / 1e9
            traffic, traffic_file = load_traffic("unfused" if args.unfused else ("dense" if args.dense else "default"))
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "avg_kernel_ms": round(avg_ms, 5),
                    "algorithmic_bytes_per_launch": algo, "traffic": traffic,
                    "traffic_source": ("%s: separate rocprofv3 --pmc passes of this command (FETCH_SIZE doubled as the "
                                       "guide prescribes); not measured in this run" % traffic_file) if traffic else None}
        pipe_bytes = B * ((ALGO_BYTES_PRE + ALGO_BYTES_DECODE) if not args.unfused
                          else (ALGO_BYTES_PRE + 3 * ALGO_BYTES_DECODE))
        result = {
            "metric": "post/pre-process images/sec (decode+NMS)", "value": round(ips, 1), "unit": "images/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "YOLOv3 416x416 80 classes, batch %d per GPU, synthetic raw heads + uint8 images; "
                                   "pre-process + %s decode + class-aware NMS; score_thr=%g iou_thr=%g%s"
                                   % (B, "unfused (API-boundary)" if args.unfused else
                                      ("fused head, threshold-first" if args.threshold_first else "fused head"), args.score_thr,
                                      args.iou_thr, "; RCCL all-gather of detections" if args.allgather else ""),
                       "global_batch": B * n_gpus, "img_size": S, "classes": C, "rows_per_image": R,
                       "candidates_per_batch": cand, "detections_per_batch": n_det,
                       "parallelism": "dp%d (batch-sharded, no data-path collective%s)"
                                      % (n_gpus, "; all-gather of results" if args.allgather else ""),
                       "batches_in_flight": 1 if args.allgather else args.depth,
                       "hip_streams": 2 if two_streams else max(1, args.batch_streams)},
            "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else "single process",
            "rccl_ranks": world if (world > 1 and backend == "nccl") else 0,
            "rank_devices": rank_devices,
            "allgather_ms_per_step": round(ag_ms_per_step, 4) if args.allgather else None,
            "per_rank_images_per_s": [round(B * args.steps / t_, 1) for t_ in rank_dt],
            "step_ms_min_max_over_ranks": [round(min(rank_dt) / args.steps * 1e3, 4), round(max(rank_dt) / args.steps * 1e3, 4)],
            "timed_region_ms": round(dt * 1e3, 2),
            "spread": {"repeats": len(job_dt), "median_ms_per_step": round(dt / args.steps * 1e3, 4),
                       "min_ms_per_step": round(job_dt[0] / args.steps * 1e3, 4),
                       "max_ms_per_step": round(job_dt[-1] / args.steps * 1e3, 4),
                       "note": "every repeat = exactly --steps steps between two barriers; value = the median repeat"},
            "pipeline_hbm": {"algorithmic_bytes_per_step": pipe_bytes,
                             "achieved_GBps_per_gpu": round(pipe_bytes * args.steps / dt / 1e9, 1),
                             "frac_of_8TBps": round(pipe_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels_ms": {k: round(v["avg_ms"], 5) for k, v in kern_all.items()},
        }
        if variants:
            result["variants"] = variants
        if roof is not None:
            result["roofline"] = roof
        if n_gpus == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.score_thr, args.iou_thr)
            result["cpu_baseline"] = cb
            result["speedup_vs_cpu_port"] = round(ips / cb["value"], 1)
        if n_gpus == 1 and args.configs != "none" and not (args.unfused or args.allgather):
            # the other BASELINE configurations, each measured here with its own roofline / NMS rate / CPU baseline
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            names = list(CONFIG_NAMES) if args.configs == "all" else [n for n in args.configs.split(",") if n]
            result["configs"] = bench_configs.run_configs(names, cpu=not args.no_cpu_baseline)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
