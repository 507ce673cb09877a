"""The build's own minimal caller of the hot path (tests, bench, examples).

Vortex's `BasePredictor` (reference predictor/base_module.py:9-113) is NOT re-implemented here: it is plain Python
that composes whatever pre-process / network / post-process modules it is given, so it runs unchanged on top of
`vortex_amd.Normalize`, `vortex_amd.YoloV3PostProcess`, ....  What this file offers instead:

  Pipeline              pre-process -> network -> post-process as one nn.Module, nothing else; `detect()` adds the
                        on-device result formatting of vortex_amd.runtime (output_format columns, pixel scaling)
  create_predictor      builds a Pipeline from a dict / namespace of components (the shape of the reference's
                        `create_model_components` result, models/detection/yolov3.py:94-112)
  get_prediction_results  host-side column slicing with the contract of reference predictor/utils.py:11-33
                        (np.take per output_format key, None for an empty result)

`BasePredictor` is kept as an alias of Pipeline so existing imports keep working."""
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn


class Pipeline(nn.Module):
    def __init__(self, model: nn.Module, preprocess: Optional[nn.Module] = None, postprocess: Optional[nn.Module] = None):
        super().__init__()
        self.preprocess = preprocess if preprocess is not None else nn.Identity()
        self.model = model
        self.postprocess = postprocess
        # the column map of the result rows travels with the network (reference models/detection/yolov3.py:56-60)
        self.output_format = dict(getattr(model, "output_format", {}))
        self.task = getattr(model, "task", "detection")
        self.eval()

    @torch.no_grad()
    def forward(self, input: torch.Tensor, *args, **kwargs):
        x = self.model(self.preprocess(input))
        return x if self.postprocess is None else self.postprocess(x, *args, **kwargs)

    def detect(self, input: torch.Tensor, *args, image_sizes: Optional[Sequence] = None, **kwargs) -> List[Dict[str, np.ndarray]]:
        """forward() + the reference's result dictionaries; with `image_sizes` (images or (h, w) sizes) the coordinates come back in
        pixels (InferenceHelper.adjust_coordinates).  Formatting runs on the device when the post-process offers the
        unpacked path (vortex_amd.runtime.format_detections), else on the host."""
        from . import runtime
        x = self.model(self.preprocess(input))
        if hasattr(self.postprocess, "forward_unpacked") and image_sizes is not None:
            out_det, keep_counts, meta = self.postprocess.forward_unpacked(x, *args, **kwargs)
            return runtime.format_detections(out_det, keep_counts, self.output_format, batch_vis=image_sizes,
                                             coordinate_fmt="relative", status=meta[-1:])
        return get_prediction_results(self.postprocess(x, *args, **kwargs), self.output_format)


BasePredictor = Pipeline


def create_predictor(model_components, **kwargs) -> Pipeline:
    def pick(name):
        if isinstance(model_components, dict):
            return model_components.get(name)
        return getattr(model_components, name, None)
    return Pipeline(pick("network"), preprocess=pick("preprocess"), postprocess=pick("postprocess"))


def _columns(spec) -> List[int]:
    ind = spec["indices"]
    return list(range(ind["start"], ind["end"])) if isinstance(ind, dict) else list(ind)


def get_prediction_results(results, output_format: Dict[str, dict]) -> List[Dict[str, Optional[np.ndarray]]]:
    plan = [(name, _columns(spec), spec["axis"]) for name, spec in sorted(output_format.items())]
    out = []
    for res in results:
        arr = res.detach().cpu().numpy() if torch.is_tensor(res) else np.asarray(res)
        row = OrderedDict()
        for name, cols, axis in plan:
            row[name] = None if arr.size == 0 else np.take(arr, cols, axis=axis)
        out.append(row)
    return out
