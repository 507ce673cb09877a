"""Runtime-side callers of the hot path with the result formatting done on the device:

  BaseRuntime.resize_batch / resize_stretch      src/runtime/vortex/runtime/basic_runtime.py:66-90
  BaseRuntime.__call__ result dicts              basic_runtime.py:103-111
  get_prediction_results                         src/development/vortex/development/predictor/utils.py:11-33
  InferenceHelper.run_inference                  src/runtime/vortex/runtime/helper.py:226-246
  InferenceHelper.adjust_coordinates             helper.py:277-310

The reference returns, per image, an OrderedDict {field: np.take(result, indices, axis)} (None when the
image has no detection) and scales bounding boxes / landmarks to pixels in place on the host.  Here the
column gather, the scaling and the packing of all images are one kernel (vx_format_results); the host
receives the counts (one small D2H) and one dense f32[total, n_cols] array (one D2H) and only slices
numpy views."""
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

_SCALED_FIELDS = ("bounding_box", "landmarks")          # helper.py:295-306


class ResultFormat:
    """An `output_format` ({field: {'indices': [...] | {'start','end'}, 'axis': 1}}) compiled to the column
    list of vx_format_results.  Fields are emitted in sorted-key order (basic_runtime.py:43, utils.py:19)."""

    def __init__(self, output_format: Dict[str, dict]):
        self.fields = sorted(output_format.keys())
        self.cols: List[int] = []
        self.kinds: List[int] = []
        self.slices: Dict[str, Tuple[int, int]] = {}
        for key in self.fields:
            spec = output_format[key]
            ind = spec["indices"]
            if isinstance(ind, dict):
                ind = list(range(ind["start"], ind["end"]))
            ind = [int(i) for i in np.asarray(ind).reshape(-1)]
            if int(spec["axis"]) != 1:
                raise RuntimeError("device result formatting handles per-row fields (axis 1) only, got axis %s for %s"
                                   % (spec["axis"], key))
            lo = len(self.cols)
            self.cols += ind
            # [..., 0::2] *= im_w ; [..., 1::2] *= im_h  (position inside the field, helper.py:298-305)
            self.kinds += [(1 + (p & 1)) if key in _SCALED_FIELDS else 0 for p in range(len(ind))]
            self.slices[key] = (lo, len(self.cols))


def _image_wh(batch_vis: Sequence, B: int, device) -> torch.Tensor:
    """(w,h) per image from HWC arrays / (h,w) tuples; images beyond len(batch_vis) (batch padding) get 1."""
    wh = np.ones((B, 2), np.float32)
    for i, v in enumerate(batch_vis):
        h, w = (v.shape[0], v.shape[1]) if hasattr(v, "shape") else (v[0], v[1])
        wh[i] = (w, h)
    return torch.from_numpy(wh).to(device, non_blocking=True)


def format_detections(out_det: torch.Tensor, keep_counts: torch.Tensor, output_format,
                      batch_vis: Optional[Sequence] = None, coordinate_fmt: Optional[str] = None,
                      status: Optional[torch.Tensor] = None) -> List[Optional[OrderedDict]]:
    """out_det f32[B,cap,D] + keep_counts i32[B] (device) -> the reference's list of OrderedDict of numpy
    arrays.  coordinate_fmt: None = no scaling; 'relative' = scale bounding_box / landmarks by the size of
    the matching image in batch_vis, 'absolute' = leave (adjust_coordinates' meaning, helper.py:308-310)."""
    fmt = output_format if isinstance(output_format, ResultFormat) else ResultFormat(output_format)
    B = out_det.size(0)
    wh = None
    if coordinate_fmt is not None:
        known_coordinate_fmt = ['relative', 'absolute']
        assert coordinate_fmt in known_coordinate_fmt, f"available 'output_coordinate_format': {known_coordinate_fmt}"
        if coordinate_fmt == 'relative':
            assert batch_vis is not None, "batch_vis (images or (h,w) sizes) is needed to scale coordinates"
            wh = _image_wh(batch_vis, B, out_det.device)
    rows = ops.format_results(out_det, keep_counts, fmt.cols, fmt.kinds if wh is not None else None, wh)
    tail = keep_counts if status is None else torch.cat((keep_counts, status.reshape(1).to(keep_counts.dtype)))
    host_counts = tail.cpu().tolist()                           # D2H 1 (sync): per-image row counts (+status)
    if status is not None:
        ops.raise_on_status(host_counts[-1])
    counts = [min(c, out_det.size(1)) for c in host_counts[:B]]
    total = sum(counts)
    host = rows[:total].cpu().numpy() if total else np.zeros((0, len(fmt.cols)), np.float32)   # D2H 2
    per_field = {k: np.ascontiguousarray(host[:, lo:hi]) for k, (lo, hi) in fmt.slices.items()}
    results, off = [], 0
    for n in counts:
        d = OrderedDict()
        for k in fmt.fields:
            d[k] = per_field[k][off:off + n] if n else None
        results.append(d)
        off += n
    return results


def resize_batch(images: List[np.ndarray], size: Tuple[int, int, int, int], resize_kind: str = 'stretch',
                 device="cuda:0") -> torch.Tensor:
    """BaseRuntime.resize_batch on the device: list of u8[Hi,Wi,3] -> u8[n,h,w,3] tensor, zero images as
    batch padding.  Keeps the reference's (h,w)-as-dsize quirk (a resized image has w rows, h columns;
    identical for square inputs, np.stack's ValueError for h != w with padding); resize_kind 'letterbox'
    is the build-side extension and resizes to (h,w) proper."""
    assert resize_kind in ['stretch', 'letterbox'] and len(size) == 4
    n, h, w, c = size if size[-1] == 3 else tuple(size[i] for i in [0, 3, 1, 2])
    assert c == 3, "device resize handles 3-channel images"
    assert len(images) <= n, "expects 'images' <= n batch ({}) got {}".format(n, len(images))
    oh, ow = (h, w) if resize_kind == 'letterbox' else (w, h)
    if (oh, ow) != (h, w) and len(images) < n:
        raise ValueError("all input arrays must have the same shape")       # np.stack, basic_runtime.py:90
    dev = torch.device(device)
    out = torch.zeros((n, oh, ow, 3), dtype=torch.uint8, device=dev)
    for i, im in enumerate(images):
        assert im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8
        src = torch.from_numpy(np.ascontiguousarray(im)).to(dev, non_blocking=True)
        ops.resize_u8(src[None], (oh, ow), letterbox=(resize_kind == 'letterbox'), out=out[i:i + 1])
    return out


def adjust_coordinates(batch_vis, batch_results, coordinate_fmt='relative'):
    """InferenceHelper.adjust_coordinates (helper.py:277-310) for results that were NOT scaled on the device."""
    known_coordinate_fmt = ['relative', 'absolute']
    assert coordinate_fmt in known_coordinate_fmt, f"available 'output_coordinate_format': {known_coordinate_fmt}"
    if coordinate_fmt != 'relative':
        return batch_results
    for vis, result in zip(batch_vis, batch_results):
        im_h, im_w = vis.shape[0], vis.shape[1]
        for key in _SCALED_FIELDS:
            if key in result and result[key] is not None:
                result[key][..., 0::2] *= im_w
                result[key][..., 1::2] *= im_h
    return batch_results


class InferenceHelper:
    """Load-free part of the reference's InferenceHelper (helper.py:208-310) over a BasePredictor living on
    an MI355X: resize on the device, run pre-process -> model -> post-process, format on the device."""

    def __init__(self, predictor, input_shape: Tuple[int, int, int, int], class_names=None, resize_kind='stretch'):
        self.model = predictor
        self.input_shape = tuple(input_shape)
        self.class_names = class_names
        self.resize_kind = resize_kind
        self.format = ResultFormat(predictor.output_format)
        self.device = next((p.device for p in predictor.parameters()), None) or \
            next((b.device for b in predictor.buffers()), torch.device("cuda:0"))

    @torch.no_grad()
    def run_inference(self, batch_imgs: List[np.ndarray], coordinate_fmt: Optional[str] = None, **kwargs):
        n = self.input_shape[0]
        assert len(batch_imgs) <= n, "expects 'images' <= n batch ({}) got {}".format(n, len(batch_imgs))
        x = resize_batch(batch_imgs, self.input_shape, self.resize_kind, self.device)
        x = self.model.model(self.model.preprocess(x))
        post = self.model.postprocess
        if not hasattr(post, "forward_unpacked"):
            raise RuntimeError("InferenceHelper needs a post-process with forward_unpacked (BatchedNMSPostProcess family)")
        out_det, keep_counts, meta = post.forward_unpacked(x, kwargs["score_threshold"], kwargs["iou_threshold"])
        return format_detections(out_det, keep_counts, self.format, batch_imgs, coordinate_fmt, status=meta[-1])

    adjust_coordinates = staticmethod(adjust_coordinates)
    resize_batch = staticmethod(resize_batch)
