"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU ops + the C greedy NMS in nms_oracle.c) of the
reference's detection pre/post-process path.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; nothing under
vortex_amd/ does, and the product path fails loudly without its HIP library.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/development/vortex/development/ unless stated otherwise).

Pinning:
  * normalize / yolo_layer_eval / yolo_decoder / the wrapper logic of batched_nms
    and BatchedNMSPostProcess are checked against golden vectors produced by the
    imported reference modules (tests/golden/gen_golden.py -> tests/golden/*.npz).
  * The greedy NMS core is third-party (torchvision, not in the image):
    PARITY UNPINNED, see nms_oracle.c.
  * prior_box_decode and resize_bilinear_u8 have no reference source in the
    snapshot: build-authored spec, PARITY UNPINNED.
"""
import ctypes
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COCO_ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45),
                (59, 119), (116, 90), (156, 198), (373, 326)]   # networks/models/detection/yolov3.py:26-29
ANCHOR_MASKS = [(6, 7, 8), (3, 4, 5), (0, 1, 2)]                 # networks/modules/heads/detection/yolov3.py:319-343


def build(force: bool = False) -> str:
    """Compile nms_oracle.c with gcc (the checker; not shipped)."""
    so = os.path.join(_HERE, "libvx_oracle.so")
    src = os.path.join(_HERE, "nms_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvx_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvx_oracle.so")
        if not os.path.exists(so):
            build()
        lib = ctypes.CDLL(so)
        lib.vx_oracle_nms.restype = ctypes.c_int64
        lib.vx_oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_float, ctypes.c_void_p]
        lib.vx_oracle_intersection.restype = ctypes.c_float
        lib.vx_oracle_intersection.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.vx_oracle_union.restype = ctypes.c_float
        lib.vx_oracle_union.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


# --------------------------------------------------------------------------- a2
def normalize(img: torch.Tensor, mean: Sequence[float], std: Sequence[float],
              scaler: int = 255, flip: bool = False) -> torch.Tensor:
    """networks/modules/preprocess/normalizer.py:13-31 (to_tensor), :34-60 (normalize),
    :74-77 (Normalize.forward), :91-95 (FlipNormalize.forward), :5-11 (flip).
    ((float(u8) / scaler) - mean[c]) / std[c], NHWC -> NCHW.  Values as the
    reference; returned contiguous (the reference returns a permuted view)."""
    assert img.dim() == 4 and img.size(3) in (1, 3)
    mean_t = torch.as_tensor(mean, dtype=torch.float)
    std_t = torch.as_tensor(std, dtype=torch.float)
    scaler_t = torch.as_tensor(scaler, dtype=torch.int)
    if flip:
        assert img.size(3) == 3
        img = img.flip(-1)
    x = img.permute(0, 3, 1, 2)
    x = x.float().div(scaler_t)
    x.sub_(mean_t[None, :, None, None]).div_(std_t[None, :, None, None])
    return x.contiguous()


# --------------------------------------------------------------------------- a3 / a4
def yolo_feature_maps(img_size: int) -> List[int]:
    """networks/modules/utils/darknet.py:136-143 — grid sizes S/8, S/16, S/32."""
    return [img_size // 8, img_size // 16, img_size // 32]


def yolo_layer_eval(pred: torch.Tensor, img_size: int, grid: int,
                    mask: Sequence[int], anchors=COCO_ANCHORS) -> torch.Tensor:
    """networks/modules/heads/detection/yolov3.py:263-304 eval branch, buffers :202-247.
    pred f32[B, 3*(5+C), G, G] -> f32[B, 3*G*G, 5+C]; rows anchor-major, then y, then x."""
    B, ch, gy, gx = pred.shape
    na = len(mask)
    npred = ch // na
    assert gy == grid and gx == grid
    stride_i = img_size // grid                                   # :207
    stride = torch.tensor([img_size / max(grid, grid)])           # :236 (fp32 tensor)
    anc = torch.Tensor([anchors[i] for i in mask])                # :216
    anchor_wh = (anc / stride).view(1, na, 1, 1, 2)               # :244-245
    yv, xv = torch.meshgrid([torch.arange(grid).float(), torch.arange(grid).float()],
                            indexing="ij")                        # :239-240
    grid_xy = torch.stack((xv, yv), 2).view(1, 1, grid, grid, 2)  # :241
    p = pred.view(B, na, npred, grid, grid).permute(0, 1, 3, 4, 2).contiguous()   # :274-279
    xy = torch.sigmoid(p[..., 0:2])                               # :281
    wh = p[..., 2:4]
    conf = torch.sigmoid(p[..., 4]).view(B, na, grid, grid, 1)    # :283-284
    cls = torch.sigmoid(p[..., 5:])                               # :285
    xy = xy + grid_xy                                             # :295
    wh = torch.exp(wh) * anchor_wh                                # :296
    bboxes = torch.cat((xy, wh), 4)
    out = torch.cat((bboxes * stride, conf, cls), -1).view(B, -1, npred)   # :298-302
    del stride_i
    return out


def yolo_head_eval(heads: Sequence[torch.Tensor], img_size: int,
                   anchors=COCO_ANCHORS) -> torch.Tensor:
    """heads = raw conv outputs for G = S/32, S/16, S/8 (coarse -> fine), as
    YoloV3Head.forward returns h1,h2,h3 (heads/detection/yolov3.py:369-376) and
    YoloV3.forward concatenates them (networks/models/detection/yolov3.py:72-76)."""
    outs = []
    for h, mask in zip(heads, ANCHOR_MASKS):
        outs.append(yolo_layer_eval(h, img_size, h.shape[-1], mask, anchors))
    return torch.cat(outs, 1)


# --------------------------------------------------------------------------- a5
def yolo_decoder(pred: torch.Tensor, score_threshold: torch.Tensor, img_size: int,
                 threshold: bool = True):
    """networks/modules/postprocess/yolov3.py:29-68 (YoloV3Decoder.forward) and :9-17
    (yolo2xywh).  pred f32[1,R,5+C] -> bboxes[K,4], scores[K], class_indexes[K] i64,
    detections[1,K,6]."""
    if torch.any(torch.isnan(pred)):                              # :35-36
        raise ValueError("your predictions have nan! please check")
    if pred.dim() != 3:                                           # :37-39
        raise RuntimeError("this routine expects predictions is a 3-dimensional tensor! "
                           "got %s dimension" % pred.dim())
    class_conf, class_pred = pred[..., 5:].max(2, keepdim=True)   # :41
    if threshold:
        sel = (pred[..., 4] > score_threshold).squeeze(0)         # :44-45
        idx = torch.nonzero(sel, as_tuple=False).squeeze(1)       # :46
        pred = pred.index_select(1, idx)                          # :47-49
        class_conf = class_conf.index_select(1, idx)
        class_pred = class_pred.index_select(1, idx)
    b = pred[..., :4] / img_size                                  # :63
    obj = pred[..., 4]
    scores = obj * class_conf.squeeze(2)                          # :65
    x1 = b[..., 0] - b[..., 2] / 2                                # :13-16
    y1 = b[..., 1] - b[..., 3] / 2
    x2 = b[..., 0] + b[..., 2] / 2
    y2 = b[..., 1] + b[..., 3] / 2
    bboxes = torch.stack((x1, y1, x2, y2), 2)
    det = torch.cat((bboxes, class_conf, class_pred.float()), 2)  # :67
    return bboxes.squeeze(0), scores.squeeze(0), class_pred.squeeze(0).squeeze(1), det


# --------------------------------------------------------------------------- a6
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold) -> torch.Tensor:
    """Stand-in for torchvision.ops.nms (reference call sites postprocess/utils/nms.py:52,:97).
    See nms_oracle.c — PARITY UNPINNED at this boundary."""
    n = int(boxes.shape[0])
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    keep = np.empty(n, dtype=np.int64)
    thr = float(iou_threshold)   # Tensor[1] -> python float, as torchvision's `float` arg does
    nk = _lib().vx_oracle_nms(b.ctypes.data, s.ctypes.data, n, thr, keep.ctypes.data)
    return torch.from_numpy(keep[:nk].copy())


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor,
                iou_threshold) -> torch.Tensor:
    """postprocess/utils/nms.py:57-98.  offsets = idxs*(2048+1) (int64, :94-95);
    boxes + offsets[:,None] promotes to fp32 (:96) — coordinates are quantised to
    ulp(2049*cls) before the IoU arithmetic."""
    if boxes.numel() == 0:                                        # :86-87
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = 2048
    offsets = idxs * (max_coordinate + 1)
    boxes_for_nms = boxes + offsets[:, None]
    return nms(boxes_for_nms, scores, iou_threshold)              # :97


def batched_nms_module(detections, class_indexes, bboxes, scores, iou_threshold,
                       top_k: Optional[int] = None) -> torch.Tensor:
    """postprocess/utils/nms.py:105-123 (BatchedNMS.forward): rank checks, then
    detections.index_select(1, keep).  top_k is a build-side extension (SURVEY.md
    §8d): first k entries of the score-descending keep list; None == reference."""
    if class_indexes.dim() != 1:
        raise RuntimeError("expects `class_indices` to be 1-dimensional tensor")
    if scores.dim() != 1:
        raise RuntimeError("expects `scores` to be 1-dimensional tensor")
    if bboxes.dim() != 2:
        raise RuntimeError("expects `bboxes` to be 2-dimensional tensor")
    if detections.dim() != 3:
        raise RuntimeError("expects `detections` to be 3-dimensional tensor")
    if detections.size(0) > 1:
        raise RuntimeError("current version only support single batch")
    keep = batched_nms(bboxes, scores, class_indexes, iou_threshold)
    if top_k is not None:
        keep = keep[:top_k]
    return detections.index_select(1, keep)


# --------------------------------------------------------------------------- a8
def yolo_postprocess(pred: torch.Tensor, score_threshold: torch.Tensor,
                     iou_threshold: torch.Tensor, img_size: int, threshold: bool = True,
                     top_k: Optional[int] = None, return_keep: bool = False):
    """postprocess/base_postprocess.py:110-138 (BatchedNMSPostProcess.forward: Python loop
    over images, decoder -> nms, squeeze(0), tuple) with postprocess/yolov3.py:71-79."""
    results, keeps = [], []
    for i in range(pred.size(0)):
        bboxes, scores, cls, det = yolo_decoder(pred[i].unsqueeze(0), score_threshold,
                                                img_size, threshold)
        keep = batched_nms(bboxes, scores, cls, iou_threshold)
        if top_k is not None:
            keep = keep[:top_k]
        keeps.append(keep)
        results.append(det.index_select(1, keep).squeeze(0))
    if return_keep:
        return tuple(results), keeps
    return tuple(results)


# --------------------------------------------------------------------------- a9 (build-authored spec)
def make_priors(img_size: int = 640, strides=(8, 16, 32), aspect_ratios=(1, 2, 3),
                scales_per_cell: int = 2) -> torch.Tensor:
    """Normalised (cx,cy,w,h) priors.  NO reference source; the only
    pinned facts are P = 50400 priors x 4 at 640 (scripts/export_check/README.md:107)
    and aspect_ratios [1,2,3] (experiments/configs/*_fpn_ssd_*.yml:16-29).  6 priors per
    cell on strides 8/16/32 give 6*(80^2+40^2+20^2) = 50400."""
    out = []
    for s in strides:
        g = img_size // s
        ys, xs = torch.meshgrid(torch.arange(g).float(), torch.arange(g).float(), indexing="ij")
        cx = ((xs + 0.5) * s / img_size).reshape(-1, 1)
        cy = ((ys + 0.5) * s / img_size).reshape(-1, 1)
        per = []
        for k in range(scales_per_cell):
            base = s * 4 * (2 ** (k / scales_per_cell)) / img_size
            for ar in aspect_ratios:
                w = base * math.sqrt(ar)
                h = base / math.sqrt(ar)
                per.append(torch.cat((cx, cy, torch.full_like(cx, w), torch.full_like(cx, h)), 1))
        out.append(torch.stack(per, 1).reshape(-1, 4))
    return torch.cat(out, 0).float().contiguous()


def prior_box_decode(inp: torch.Tensor, priors: torch.Tensor, score_threshold: torch.Tensor,
                     n_landmarks: int = 0, variances=(0.1, 0.2), threshold: bool = True):
    """Prior-box decoder for FPN-SSD / RetinaFace heads.  BUILD-AUTHORED SPEC — the
    reference's postprocess/retinaface.py and fpn_ssd sources are absent from the snapshot;
    evidence for the contract: scripts/export_check/README.md:54-63 (batch 1;
    input.size(2) == 6 + 2*n_landmarks for RetinaFace; score_threshold.size()==[1];
    loc.size(1)==4; priors.size(1)==4) and :113 (output row = 16 floats).  PARITY UNPINNED.

    inp f32[1, P, 4 + 2*L + NC] = loc(4) | landmarks(2L) | class logits (NC incl. background 0)
      (RetinaFace: L=5, NC=2 -> 16 columns; SSD: L=0, NC=C+1).
    conf = softmax(logits); class_conf, class_pred = max over the non-background classes
    (first max wins); keep rows with class_conf > score_threshold (ascending row order);
    cxcy = p_c + loc_xy*v0*p_wh ; wh = p_wh*exp(loc_wh*v1); xyxy = cxcy -/+ wh/2;
    landmark_k = p_c + ldm_k*v0*p_wh.
    Returns bboxes[K,4], scores[K] (= class_conf), class_indexes[K] i64 (0-based foreground
    class), detections[1,K,6+2L] = xyxy | class_conf | class | landmarks."""
    assert inp.dim() == 3 and inp.size(0) == 1
    L = n_landmarks
    x = inp[0]
    loc, ldm, logits = x[:, :4], x[:, 4:4 + 2 * L], x[:, 4 + 2 * L:]
    m = logits.max(1, keepdim=True)[0]
    e = torch.exp(logits - m)
    conf = e / e.sum(1, keepdim=True)
    class_conf, class_pred = conf[:, 1:].max(1)
    v0, v1 = variances
    cxcy = priors[:, :2] + loc[:, :2] * v0 * priors[:, 2:]
    wh = priors[:, 2:] * torch.exp(loc[:, 2:] * v1)
    half = wh / 2
    bboxes = torch.cat((cxcy - half, cxcy + half), 1)
    cols = [bboxes, class_conf[:, None], class_pred.float()[:, None]]
    if L:
        pts = priors[:, None, :2] + ldm.reshape(-1, L, 2) * v0 * priors[:, None, 2:]
        cols.append(pts.reshape(-1, 2 * L))
    det = torch.cat(cols, 1)
    if threshold:
        idx = torch.nonzero(class_conf > score_threshold, as_tuple=False).squeeze(1)
        bboxes, class_conf, class_pred, det = (bboxes.index_select(0, idx), class_conf.index_select(0, idx),
                                               class_pred.index_select(0, idx), det.index_select(0, idx))
    return bboxes, class_conf, class_pred, det.unsqueeze(0)


def prior_postprocess(inp, priors, score_threshold, iou_threshold, n_landmarks=0,
                      top_k: Optional[int] = None, return_keep: bool = False):
    """decoder -> BatchedNMS per image, as base_postprocess.py:110-138."""
    results, keeps = [], []
    for i in range(inp.size(0)):
        b, s, c, det = prior_box_decode(inp[i].unsqueeze(0), priors, score_threshold, n_landmarks)
        keep = batched_nms(b, s, c, iou_threshold)
        if top_k is not None:
            keep = keep[:top_k]
        keeps.append(keep)
        results.append(det.index_select(1, keep).squeeze(0))
    if return_keep:
        return tuple(results), keeps
    return tuple(results)


# --------------------------------------------------------------------------- a1 (build-authored spec)
def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int, letterbox: bool = False,
                       pad_value: int = 0) -> np.ndarray:
    """u8[H,W,C] -> u8[out_h,out_w,C].  Stands where the reference calls cv2.resize
    (src/runtime/vortex/runtime/basic_runtime.py:66-68, default INTER_LINEAR, stretch).
    cv2 is not in the image and the reference holds no resize fixture: PARITY UNPINNED.
    Restates OpenCV's published u8 INTER_LINEAR scheme: half-pixel centres
    src = (dst+0.5)*scale-0.5, taps clamped to the border, 11-bit fixed-point weights
    (round-to-nearest of w*2048), horizontal pass in int32, vertical pass
    (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
    letterbox (build extension): scale=min(out_h/H,out_w/W), resized size
    round(H*scale) x round(W*scale), centred, pad_value elsewhere."""
    H, W, C = img.shape
    if letterbox:
        sc = min(out_h / H, out_w / W)
        nh, nw = max(1, int(round(H * sc))), max(1, int(round(W * sc)))
        top, left = (out_h - nh) // 2, (out_w - nw) // 2
        out = np.full((out_h, out_w, C), pad_value, dtype=np.uint8)
        out[top:top + nh, left:left + nw] = resize_bilinear_u8(img, nh, nw)
        return out

    def taps(n_out, n_in):
        # OpenCV resize(): inv_scale = (double)dsize/ssize, scale = 1./inv_scale; fx = (float)((dx+0.5)*scale-0.5);
        # sx = cvFloor(fx); fx -= sx; border clamps zero fx; the two weights are rounded separately:
        # saturate_cast<short>({1.f-fx, fx} * 2048) (round half to even).
        scale = np.float64(1.0) / (np.float64(n_out) / np.float64(n_in))
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int32)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0 = np.where(lo, 0, i0); fr = np.where(lo, np.float32(0), fr)
        hi = i0 >= n_in - 1
        i0 = np.where(hi, n_in - 1, i0); fr = np.where(hi, np.float32(0), fr)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int32)
        w1 = np.rint(fr * np.float32(2048)).astype(np.int32)
        return i0, i1, w0, w1

    x0, x1, a0, a1 = taps(out_w, W)
    y0, y1, b0, b1 = taps(out_h, H)
    src = img.astype(np.int32)
    hrow = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]   # [H,out_w,C]
    r0 = hrow[y0]; r1 = hrow[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_batch(images: Sequence[np.ndarray], size: Tuple[int, int, int, int],
                 letterbox: bool = False) -> np.ndarray:
    """src/runtime/vortex/runtime/basic_runtime.py:75-90: resize each image, pad the batch with zero
    images up to n, stack.  Quirk kept: the reference hands (h,w) to cv2.resize, whose
    dsize is (width,height), so a resized image has w rows and h columns -- identical for the square
    inputs every shipped config uses; for h != w the stack with (h,w,c) pad images raises ValueError
    exactly as np.stack does there.  letterbox (build extension) resizes to (h,w) proper."""
    assert len(size) == 4
    n, h, w, c = size if size[-1] == 3 else tuple(size[i] for i in [0, 3, 1, 2])
    if letterbox:
        out = [resize_bilinear_u8(im, h, w, True) for im in images]
    else:
        out = [resize_bilinear_u8(im, w, h) for im in images]
    out += [np.zeros((h, w, c), dtype=images[0].dtype)] * (n - len(images))
    return np.stack(out)


# --------------------------------------------------------------------------- a10
def get_prediction_results(results, output_format):
    """predictor/utils.py:11-33 — np.take per output_format key, None for empty results."""
    from collections import OrderedDict
    fields = sorted(output_format.keys())
    final = []
    for r in results:
        r = r.detach().cpu().numpy() if isinstance(r, torch.Tensor) else np.asarray(r)
        d = OrderedDict()
        for k in fields:
            ind = output_format[k]["indices"]
            if isinstance(ind, dict):
                ind = list(range(ind["start"], ind["end"]))
            d[k] = np.take(r, indices=ind, axis=output_format[k]["axis"]) if all(r.shape) else None
        final.append(d)
    return final


def adjust_coordinates(batch_vis, batch_results, coordinate_fmt="relative"):
    """InferenceHelper.adjust_coordinates (src/runtime/vortex/runtime/helper.py:277-310): in-place scaling
    of bounding_box / landmarks x by im_w and y by im_h when the coordinates are 'relative'."""
    assert coordinate_fmt in ["relative", "absolute"]
    if coordinate_fmt != "relative":
        return batch_results
    for vis, result in zip(batch_vis, batch_results):
        im_h, im_w, _ = vis.shape
        for key in ("bounding_box", "landmarks"):
            if key in result and result[key] is not None:
                a = result[key]
                a[..., 0::2] *= im_w
                a[..., 1::2] *= im_h
    return batch_results


# --------------------------------------------------------------------------- §8f-1 DETR
def detr_postprocess(inputs: torch.Tensor, score_threshold: torch.Tensor) -> Tuple[torch.Tensor, ...]:
    """DETRPostProcess.forward / _process_single_batch (networks/models/detection/detr.py:539-559) with
    cxcywh_to_xyxy (detr.py:932-936): softmax over all logits, max over all but the last ("no object"),
    strict threshold on that probability, rows kept in query order."""
    if inputs.ndim != 3:
        raise RuntimeError("DETR postprocess input must have dimension of 3 "
                           "(num_batch, num_queries, (num_classes + 1) + 4)")
    out = []
    for x in inputs:
        x_c, y_c, w, h = x[:, :4].unbind(-1)
        bbox = torch.stack([(x_c - 0.5 * w), (y_c - 0.5 * h), (x_c + 0.5 * w), (y_c + 0.5 * h)], dim=-1)
        probas = x[:, 4:].softmax(-1)[:, :-1]
        conf, labels = probas.max(-1, keepdim=True)
        keep = (conf > score_threshold).flatten()
        out.append(torch.cat((bbox, conf, labels.float()), -1)[keep])
    return tuple(out)


def synth_detr_outputs(B: int, Q: int = 100, n_logits: int = 92, seed: int = 0) -> torch.Tensor:
    """DETR head output [B,Q,4+n_logits]: boxes sigmoid(N(0,1)) (cx,cy,w,h in 0..1), class logits N(0,2),
    no-object logit N(2,2) (most queries are background, as after training)."""
    g = torch.Generator().manual_seed(seed)
    box = torch.sigmoid(torch.randn(B, Q, 4, generator=g))
    lg = torch.randn(B, Q, n_logits, generator=g) * 2
    lg[..., -1] += 2
    return torch.cat((box, lg), -1).contiguous()


# --------------------------------------------------------------------------- classification post-process
def softmax_postprocess(logits: torch.Tensor) -> torch.Tensor:
    """SoftmaxPostProcess.forward with dim=1, keepdim=False (networks/modules/postprocess/base_postprocess.py:
    147-150): softmax over the classes, max probability and its index -> [B,2] = (label as float, confidence)."""
    p = torch.softmax(logits.float(), dim=1)
    conf, label = p.max(dim=1)
    return torch.stack((label.float(), conf), dim=1)


# --------------------------------------------------------------------------- synthetic inputs
def synth_images(B: int, S: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)


def synth_yolo_heads(B: int, S: int = 416, C: int = 80, seed: int = 0) -> List[torch.Tensor]:
    """Raw YOLO head tensors, coarse -> fine: t_xy~N(0,1), t_wh~N(0,0.5), t_obj~N(-4,2),
    t_cls~N(-2,1.5)."""
    g = torch.Generator().manual_seed(seed)
    heads = []
    for G in (S // 32, S // 16, S // 8):
        t = torch.randn(B, 3, 5 + C, G, G, generator=g)
        t[:, :, 2:4] *= 0.5
        t[:, :, 4] = t[:, :, 4] * 2 - 4
        t[:, :, 5:] = t[:, :, 5:] * 1.5 - 2
        heads.append(t.view(B, 3 * (5 + C), G, G).contiguous())
    return heads


def synth_prior_head(B: int, P: int, n_landmarks: int, n_logits: int, seed: int = 0) -> torch.Tensor:
    """loc~N(0,1), ldm~N(0,1), conf logits~N(-3,2) on foreground, background logit ~N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    L = n_landmarks
    x = torch.randn(B, P, 4 + 2 * L + n_logits, generator=g)
    x[..., 4 + 2 * L + 1:] = x[..., 4 + 2 * L + 1:] * 2 - 3
    return x.contiguous()


def synth_stress_boxes(K: int, C: int = 80, seed: int = 0):
    """NMS stress (BASELINE.json configs[4]): centre~U(0,1)^2, wh~U(0.02,0.3)^2,
    score~U(0,1), class~U{0..C-1}."""
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(K, 2, generator=g)
    wh = torch.rand(K, 2, generator=g) * 0.28 + 0.02
    boxes = torch.cat((c - wh / 2, c + wh / 2), 1).contiguous()
    scores = torch.rand(K, generator=g)
    cls = torch.randint(0, C, (K,), generator=g, dtype=torch.int64)
    return boxes, scores, cls
