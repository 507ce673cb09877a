"""Functional layer: torch tensors in, libvortex_hip.so calls, torch tensors out.

Every function launches on torch's current stream, allocates its outputs/workspace with
torch (device memory plumbing) and performs NO host synchronisation; the nn.Module layer
(preprocess.py / heads.py / postprocess.py / prior.py) does the single D2H read of the
counts + status words where the reference's API needs Python-visible shapes."""
import contextlib
import functools
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

COCO_ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45),
                (59, 119), (116, 90), (156, 198), (373, 326)]   # reference networks/models/detection/yolov3.py:26-29


NMS_SINGLE_LAUNCH_MAX = 16384      # candidates per image the single-launch NMS kernel sorts in LDS (VX_SORT_CHMAX)


def yolo_rows(img_size: int) -> int:
    return 3 * sum((img_size // d) ** 2 for d in (32, 16, 8))


def _on_device(fn):
    """The library keys per-device state (LDS attributes, events, the static scratch) off HIP's CURRENT device and
    launches on the stream it is given: when the tensors live on another device than the current one, make theirs
    current for the duration of the call (a torch op would do the same).  One comparison when nothing has to change."""
    @functools.wraps(fn)
    def call(*args, **kwargs):
        t = args[0]
        if isinstance(t, (list, tuple)):
            t = t[0]
        if torch.is_tensor(t) and t.is_cuda and t.device.index not in (None, torch.cuda.current_device()):
            with torch.cuda.device(t.device.index):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return call


def _ws(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator returns >=512-byte aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------------------- K1
@_on_device
def preprocess(img: torch.Tensor, mean: Sequence[float], std: Sequence[float], scaler: int = 255,
               flip: bool = False, out_size: Optional[Tuple[int, int]] = None,
               letterbox: bool = False, pad_value: int = 0) -> torch.Tensor:
    """u8[B,H,W,3] -> f32[B,3,Hd,Wd]; see vx_preprocess_u8hwc_f32nchw."""
    L.require_cuda(img, "input")
    assert img.dtype == torch.uint8 and img.dim() == 4 and img.size(3) == 3
    img = img.contiguous()
    B, Hs, Ws, _ = img.shape
    Hd, Wd = (Hs, Ws) if out_size is None else (int(out_size[0]), int(out_size[1]))
    out = torch.empty((B, 3, Hd, Wd), dtype=torch.float32, device=img.device)
    rc = L.lib().vx_preprocess_u8hwc_f32nchw(
        img.data_ptr(), out.data_ptr(), B, Hs, Ws, Hd, Wd, L.f3(mean), L.f3(std), int(scaler),
        int(bool(flip)), L.VX_RESIZE_LETTERBOX if letterbox else L.VX_RESIZE_STRETCH, int(pad_value),
        L.stream_ptr(img.device))
    L.check(rc, "vx_preprocess_u8hwc_f32nchw")
    return out


@_on_device
def resize_u8(img: torch.Tensor, out_size: Tuple[int, int], letterbox: bool = False, pad_value: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """u8[B,Hs,Ws,3] -> u8[B,Hd,Wd,3] (cv2.resize INTER_LINEAR scheme); see vx_resize_u8hwc."""
    L.require_cuda(img, "input")
    assert img.dtype == torch.uint8 and img.dim() == 4 and img.size(3) == 3
    img = img.contiguous()
    B, Hs, Ws, _ = img.shape
    Hd, Wd = int(out_size[0]), int(out_size[1])
    if out is None:
        out = torch.empty((B, Hd, Wd, 3), dtype=torch.uint8, device=img.device)
    assert out.is_contiguous() and tuple(out.shape) == (B, Hd, Wd, 3) and out.dtype == torch.uint8
    rc = L.lib().vx_resize_u8hwc(img.data_ptr(), out.data_ptr(), B, Hs, Ws, Hd, Wd,
                                 L.VX_RESIZE_LETTERBOX if letterbox else L.VX_RESIZE_STRETCH, int(pad_value),
                                 L.stream_ptr(img.device))
    L.check(rc, "vx_resize_u8hwc")
    return out


@_on_device
def format_results(out_det: torch.Tensor, keep_counts: torch.Tensor, cols: Sequence[int],
                   kinds: Optional[Sequence[int]] = None, image_wh: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f32[B,cap,D] + i32[B] -> f32[B*cap, len(cols)]: kept rows of all images back to back, columns
    gathered by `cols`, kind 1/2 columns scaled by image_wh[b] = (w,h); see vx_format_results."""
    import ctypes
    L.require_cuda(out_det, "detections")
    B, cap, D = out_det.shape
    n = len(cols)
    dst = torch.empty((B * cap, n), dtype=torch.float32, device=out_det.device)
    c_cols = (ctypes.c_int32 * n)(*[int(c) for c in cols])
    c_kinds = (ctypes.c_uint8 * n)(*[int(k) for k in kinds]) if kinds is not None else None
    if image_wh is not None:
        L.require_cuda(image_wh, "image_wh")
        assert image_wh.dtype == torch.float32 and tuple(image_wh.shape) == (B, 2) and image_wh.is_contiguous()
    rc = L.lib().vx_format_results(out_det.data_ptr(), keep_counts.data_ptr(), B, cap, D, c_cols, c_kinds, n,
                                   image_wh.data_ptr() if image_wh is not None else None, dst.data_ptr(), None,
                                   L.stream_ptr(out_det.device))
    L.check(rc, "vx_format_results")
    return dst


# --------------------------------------------------------------------------------------- K2
@_on_device
def yolo_head_decode(x: torch.Tensor, anchor_wh: Sequence[float], stride: float,
                     out: Optional[torch.Tensor] = None, row_offset: int = 0) -> torch.Tensor:
    """f32[B,3(5+C),G,G] -> rows [row_offset, row_offset+3GG) of out f32[B,R,5+C]."""
    L.require_cuda(x, "prediction")
    assert x.dtype == torch.float32 and x.dim() == 4 and x.size(2) == x.size(3)
    x = x.contiguous()
    B, ch, G, _ = x.shape
    assert ch % 3 == 0
    NP = ch // 3
    if out is None:
        out = torch.empty((B, 3 * G * G, NP), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.size(0) == B and out.size(2) == NP
    rc = L.lib().vx_yolo_head_decode(x.data_ptr(), out.data_ptr(), B, G, NP - 5, L.f6(anchor_wh), float(stride),
                                     out.size(1) * NP, int(row_offset), L.stream_ptr(x.device))
    L.check(rc, "vx_yolo_head_decode")
    return out


@_on_device
def yolo_head_decode3(heads: Sequence[torch.Tensor], img_size: int, anchors=None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Three raw heads (coarse -> fine, f32[B,3(5+C),G,G], G = S/32, S/16, S/8) -> f32[B,R,5+C] in ONE launch
    (vx_yolo_head_decode3): YoloV3Layer x3 + the concat of YoloV3.forward, same bits as three yolo_head_decode calls."""
    assert len(heads) == 3
    hs = []
    for h, d in zip(heads, (32, 16, 8)):
        L.require_cuda(h, "head")
        assert h.dtype == torch.float32 and h.dim() == 4 and h.size(2) == img_size // d == h.size(3)
        hs.append(h.contiguous())
    B, ch = hs[0].shape[:2]
    assert ch % 3 == 0 and all(h.size(0) == B and h.size(1) == ch for h in hs)
    NP = ch // 3
    R = yolo_rows(img_size)
    if out is None:
        out = torch.empty((B, R, NP), dtype=torch.float32, device=hs[0].device)
    assert out.is_contiguous() and tuple(out.shape) == (B, R, NP) and out.dtype == torch.float32
    if B == 0:
        return out
    anchors = COCO_ANCHORS if anchors is None else anchors
    rc = L.lib().vx_yolo_head_decode3(hs[0].data_ptr(), hs[1].data_ptr(), hs[2].data_ptr(), out.data_ptr(), B,
                                      int(img_size), NP - 5, L.f18([float(v) for a in anchors for v in a]),
                                      L.stream_ptr(hs[0].device))
    L.check(rc, "vx_yolo_head_decode3")
    return out


class Candidates:
    """Dense per-image candidate lists produced by a decoder (device buffers, capacity R)."""
    __slots__ = ("boxes", "scores", "cls", "det", "meta", "B", "R", "D")

    def __init__(self, B, R, D, device):
        self.B, self.R, self.D = B, R, D
        self.boxes = torch.empty((B, R, 4), dtype=torch.float32, device=device)
        self.scores = torch.empty((B, R), dtype=torch.float32, device=device)
        self.cls = torch.empty((B, R), dtype=torch.int64, device=device)
        self.det = torch.empty((B, R, D), dtype=torch.float32, device=device)
        # meta = counts[B] | keep_counts[B] | status[1]  (one D2H copy reads everything)
        self.meta = torch.zeros((2 * B + 1,), dtype=torch.int32, device=device)

    @property
    def counts(self):
        return self.meta[:self.B]

    @property
    def keep_counts(self):
        return self.meta[self.B:2 * self.B]

    @property
    def status(self):
        return self.meta[2 * self.B:]


# --------------------------------------------------------------------------------------- K3
@_on_device
def yolo_decode(pred: torch.Tensor, img_size: float, score_threshold: float, use_threshold: bool = True) -> Candidates:
    L.require_cuda(pred, "predictions")
    assert pred.dtype == torch.float32 and pred.dim() == 3
    pred = pred.contiguous()
    B, R, NP = pred.shape
    c = Candidates(B, R, 6, pred.device)
    nb = L.lib().vx_yolo_decode_workspace_bytes(B, R)
    ws = _ws(nb, pred.device)
    rc = L.lib().vx_yolo_decode(pred.data_ptr(), B, R, NP - 5, float(img_size), float(score_threshold),
                                int(bool(use_threshold)), c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(),
                                c.det.data_ptr(), c.counts.data_ptr(), c.status.data_ptr(), ws.data_ptr(), nb,
                                L.stream_ptr(pred.device))
    L.check(rc, "vx_yolo_decode")
    return c


# --------------------------------------------------------------------------------------- K2+K3 fused
@_on_device
def yolo_fused_decode(heads: Sequence[torch.Tensor], img_size: int, score_threshold: float,
                      use_threshold: bool = True, anchors=COCO_ANCHORS, threshold_first: bool = False) -> Candidates:
    """heads: raw conv outputs for G = S/32, S/16, S/8 (coarse -> fine).  threshold_first: read the objectness
    planes first and the other planes only for surviving rows (VX_DECODE_THRESHOLD_FIRST; same outputs, the NaN
    check then covers objectness everywhere but box/class values of surviving lanes only)."""
    assert len(heads) == 3
    hs = []
    for h, d in zip(heads, (32, 16, 8)):
        L.require_cuda(h, "head")
        assert h.dtype == torch.float32 and h.dim() == 4 and h.size(2) == img_size // d == h.size(3)
        hs.append(h.contiguous())
    B, ch = hs[0].shape[:2]
    C = ch // 3 - 5
    R = yolo_rows(img_size)
    c = Candidates(B, R, 6, hs[0].device)
    nb = L.lib().vx_yolo_fused_workspace_bytes(B, int(img_size))
    ws = _ws(nb, hs[0].device)
    flat = [float(v) for a in anchors for v in a]
    rc = L.lib().vx_yolo_fused_decode(hs[0].data_ptr(), hs[1].data_ptr(), hs[2].data_ptr(), B, int(img_size), C,
                                      L.f18(flat), float(score_threshold),
                                      int(bool(use_threshold)) | (2 if threshold_first else 0),
                                      c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(), c.det.data_ptr(),
                                      c.counts.data_ptr(), c.status.data_ptr(), ws.data_ptr(), nb,
                                      L.stream_ptr(hs[0].device))
    L.check(rc, "vx_yolo_fused_decode")
    return c


_ANCHOR_CACHE = {}


@_on_device
def yolo_fused_postprocess(heads: Sequence[torch.Tensor], img_size: int, score_threshold: float, iou_threshold: float,
                           use_threshold: bool = True, anchors=COCO_ANCHORS, threshold_first: bool = False,
                           top_k: Optional[int] = None, optimistic_nms: bool = False,
                           tail_stream: Optional["torch.cuda.Stream"] = None):
    """Raw heads -> (packed f32[B*out_cap, 6], meta i32[2B+1] on the device, pinned host mirror of meta) with ONE
    library call (vx_yolo_fused_postprocess): decode, NMS and packing are chained on the current stream; wait
    for an event recorded after this call before reading the mirror.  tail_stream: NMS + packing run there
    (vx_yolo_fused_postprocess_overlapped); record the event on THAT stream."""
    assert len(heads) == 3
    hs = []
    for h, d in zip(heads, (32, 16, 8)):
        L.require_cuda(h, "head")
        assert h.dtype == torch.float32 and h.dim() == 4 and h.size(2) == img_size // d == h.size(3)
        hs.append(h.contiguous())
    B, ch = hs[0].shape[:2]
    C = ch // 3 - 5
    R = yolo_rows(img_size)
    dev = hs[0].device
    tk = int(top_k) if top_k else 0
    out_cap = min(R, tk) if tk > 0 else R
    key = tuple(tuple(a) for a in anchors)
    f18 = _ANCHOR_CACHE.get(key)
    if f18 is None:
        f18 = _ANCHOR_CACHE[key] = L.f18([float(v) for a in anchors for v in a])
    lib = L.lib()
    nb = lib.vx_yolo_fused_postprocess_workspace_bytes(B, int(img_size), tk)
    ws = _ws(nb, dev)
    packed = torch.empty((B * out_cap, 6), dtype=torch.float32, device=dev)
    meta = torch.empty((2 * B + 1,), dtype=torch.int32, device=dev)
    from .postprocess import mirror_is_coherent
    coherent = mirror_is_coherent(dev)
    host = torch.empty((2 * B + 1,), dtype=torch.int32, pin_memory=True)
    flags = int(bool(use_threshold)) | (2 if threshold_first else 0) | (4 if optimistic_nms else 0)
    if tail_stream is None:
        rc = lib.vx_yolo_fused_postprocess(hs[0].data_ptr(), hs[1].data_ptr(), hs[2].data_ptr(), B, int(img_size), C,
                                           f18, float(score_threshold), flags, float(iou_threshold), tk,
                                           packed.data_ptr(), meta.data_ptr(), host.data_ptr() if coherent else None,
                                           ws.data_ptr(), nb, L.stream_ptr(dev))
    else:
        for t in (packed, meta, ws):
            t.record_stream(tail_stream)           # allocated for the current stream, last used on the tail stream
        rc = lib.vx_yolo_fused_postprocess_overlapped(hs[0].data_ptr(), hs[1].data_ptr(), hs[2].data_ptr(), B,
                                                      int(img_size), C, f18, float(score_threshold), flags,
                                                      float(iou_threshold), tk, packed.data_ptr(), meta.data_ptr(),
                                                      host.data_ptr() if coherent else None, ws.data_ptr(), nb,
                                                      L.stream_ptr(dev), tail_stream.cuda_stream)
    L.check(rc, "vx_yolo_fused_postprocess")
    if not coherent:                                   # kernel stores to pinned memory are not host-visible here: D2H copy
        with torch.cuda.stream(tail_stream) if tail_stream is not None else contextlib.nullcontext():
            host.copy_(meta, non_blocking=True)
    return packed, meta, host


# --------------------------------------------------------------------------------------- K4
@_on_device
def prior_decode(inp: torch.Tensor, priors: torch.Tensor, n_landmarks: int, score_threshold: float,
                 use_threshold: bool = True, variances=(0.1, 0.2)) -> Candidates:
    L.require_cuda(inp, "input")
    L.require_cuda(priors, "priors")
    assert inp.dtype == torch.float32 and inp.dim() == 3 and priors.dtype == torch.float32
    inp = inp.contiguous()
    priors = priors.contiguous()
    B, P, ncol = inp.shape
    assert priors.shape == (P, 4)
    n_logits = ncol - 4 - 2 * n_landmarks
    c = Candidates(B, P, 6 + 2 * n_landmarks, inp.device)
    nb = L.lib().vx_prior_decode_workspace_bytes(B, P, n_landmarks)
    ws = _ws(nb, inp.device)
    rc = L.lib().vx_prior_decode(inp.data_ptr(), priors.data_ptr(), B, P, n_landmarks, n_logits,
                                 float(variances[0]), float(variances[1]), float(score_threshold),
                                 int(bool(use_threshold)), c.boxes.data_ptr(), c.scores.data_ptr(), c.cls.data_ptr(),
                                 c.det.data_ptr(), c.counts.data_ptr(), c.status.data_ptr(), ws.data_ptr(), nb,
                                 L.stream_ptr(inp.device))
    L.check(rc, "vx_prior_decode")
    return c


# --------------------------------------------------------------------------------------- K5-K7
@_on_device
def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, cls: torch.Tensor, det: torch.Tensor,
                counts: Optional[torch.Tensor], iou_threshold: float, n_classes: int = 0,
                top_k: Optional[int] = None, meta: Optional[torch.Tensor] = None, optimistic: bool = False):
    """boxes f32[B,K,4], scores f32[B,K], cls i64[B,K], det f32[B,K,D], counts i32[B] or None.
    Returns keep_idx i64[B,out_cap], out_det f32[B,out_cap,D], keep_counts i32[B], status i32[1]
    (device tensors; rows beyond keep_counts[b] are undefined).
    optimistic (K > 16384 only): one launch (vx_batched_nms_small); if an image has more than 16384 candidates the
    status word carries VX_STATUS_DEFERRED and the call must be repeated without `optimistic`."""
    for t, n in ((boxes, "bboxes"), (scores, "scores"), (cls, "class_indexes"), (det, "detections")):
        L.require_cuda(t, n)
    assert boxes.dtype == torch.float32 and scores.dtype == torch.float32 and det.dtype == torch.float32
    assert cls.dtype == torch.int64
    boxes, scores, cls, det = boxes.contiguous(), scores.contiguous(), cls.contiguous(), det.contiguous()
    B, K = scores.shape
    D = det.size(2)
    assert boxes.shape == (B, K, 4) and cls.shape == (B, K) and det.shape[:2] == (B, K)
    dev = boxes.device
    tk = int(top_k) if top_k else 0
    out_cap = min(K, tk) if tk > 0 else K
    keep_idx = torch.empty((B, out_cap), dtype=torch.int64, device=dev)
    out_det = torch.empty((B, out_cap, D), dtype=torch.float32, device=dev)
    if meta is None:
        meta = torch.zeros((2 * B + 1,), dtype=torch.int32, device=dev)
    keep_counts, status = meta[B:2 * B], meta[2 * B:]
    nb = L.lib().vx_batched_nms_workspace_bytes(B, K)
    ws = _ws(nb, dev)
    fn = L.lib().vx_batched_nms_small if optimistic else L.lib().vx_batched_nms
    rc = fn(boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), det.data_ptr(),
                                counts.data_ptr() if counts is not None else None, B, K, D, int(n_classes),
                                float(iou_threshold), tk, keep_idx.data_ptr(), out_det.data_ptr(),
                                keep_counts.data_ptr(), out_cap, status.data_ptr(), ws.data_ptr(), nb,
                                L.stream_ptr(dev))
    L.check(rc, "vx_batched_nms")
    return keep_idx, out_det, keep_counts, status


def candidates_nms(c: Candidates, iou_threshold: float, n_classes: int = 0, top_k: Optional[int] = None,
                   optimistic: bool = False):
    """decoder output -> NMS, sharing the meta words so one D2H copy reads counts, keep_counts, status."""
    return batched_nms(c.boxes, c.scores, c.cls, c.det, c.counts, iou_threshold, n_classes, top_k, meta=c.meta,
                       optimistic=optimistic)


@_on_device
def pack_rows(out_det: torch.Tensor, keep_counts: torch.Tensor, meta: Optional[torch.Tensor] = None,
              host_meta: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f32[B,cap,D] + i32[B] -> f32[B*cap, D] with the kept rows of all images back to back.  With `meta`
    (device i32[n]) and `host_meta` (pinned host i32[n]) the kernel also mirrors the meta words into the pinned
    buffer, so the caller only waits for an event instead of issuing a D2H copy."""
    B, cap, D = out_det.shape
    packed = torch.empty((B * cap, D), dtype=torch.float32, device=out_det.device)
    src = dst = None
    n = 0
    if host_meta is not None:
        assert meta is not None and host_meta.is_pinned() and host_meta.dtype == torch.int32 == meta.dtype
        assert host_meta.numel() == meta.numel() and meta.is_contiguous()
        src, dst, n = meta.data_ptr(), host_meta.data_ptr(), meta.numel()
    rc = L.lib().vx_pack_rows(out_det.data_ptr(), keep_counts.data_ptr(), B, cap, D, packed.data_ptr(), None,
                              src, dst, n, L.stream_ptr(out_det.device))
    L.check(rc, "vx_pack_rows")
    return packed


@_on_device
def detr_decode(inputs: torch.Tensor, score_threshold: float):
    """f32[B,Q,4+n_logits] -> (rows f32[B,Q,6] compacted per image in query order, meta i32[2B+1] with the
    per-image row counts at [B:2B], same layout as the NMS paths so finish_results() applies)."""
    L.require_cuda(inputs, "inputs")
    x = inputs.contiguous().float()
    B, Q, ncol = x.shape
    out = torch.empty((B, Q, 6), dtype=torch.float32, device=x.device)
    meta = torch.zeros(2 * B + 1, dtype=torch.int32, device=x.device)
    rc = L.lib().vx_detr_decode(x.data_ptr(), B, Q, ncol - 4, float(score_threshold), out.data_ptr(),
                                meta[B:2 * B].data_ptr(), L.stream_ptr(x.device))
    L.check(rc, "vx_detr_decode")
    return out, meta


@_on_device
def softmax_max(logits: torch.Tensor) -> torch.Tensor:
    """f32[B,NC] logits -> f32[B,2] = (float(first arg-max of softmax), max probability) (vx_softmax_max)."""
    L.require_cuda(logits, "input")
    x = logits.contiguous().float()
    B, NC = x.shape
    out = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    if B:
        rc = L.lib().vx_softmax_max(x.data_ptr(), B, NC, out.data_ptr(), L.stream_ptr(x.device))
        L.check(rc, "vx_softmax_max")
    return out


def status_ok(status: int) -> bool:
    """True when the status word reports nothing a caller must act on.  The two coordinate-range bits are only a
    refusal together (VX_STATUS_RANGE / VX_STATUS_RANGE2 in include/vortex_hip.h): normalised boxes that stick out of
    the image set RANGE2 alone, pixel boxes beyond 1024 set RANGE alone -- both are fine."""
    both = L.VX_STATUS_RANGE | L.VX_STATUS_RANGE2
    rest = status & ~both
    return rest == 0 and (status & both) != both


def raise_on_status(status: int):
    """Map device status bits to the reference's exceptions (VX_STATUS_DEFERRED is not an error: see batched_nms)."""
    if status & L.VX_STATUS_NAN:
        # reference: networks/modules/postprocess/yolov3.py:35-36
        raise ValueError("your predictions have nan! please check")
    if status & L.VX_STATUS_BAD_CLASS:
        raise L.VortexHipError("class index outside [0, %d): not supported by libvortex_hip" % L.VX_MAX_CLASSES)
    if (status & L.VX_STATUS_RANGE) and (status & L.VX_STATUS_RANGE2):
        raise L.VortexHipError("class-aware NMS needs every box coordinate inside (-1024, 1024) (normalised boxes) or every "
                               "one inside [0, 2048] (pixel boxes up to the reference's max_coordinate): beyond that the "
                               "reference's 2049*class shift lets classes overlap, which the per-class sweep of "
                               "libvortex_hip does not reproduce")
