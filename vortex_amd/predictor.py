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

`BasePredictor` keeps the reference's constructor contract on top of Pipeline (see the class), so code that imports
the name gets the reference's checks and defaults, not a look-alike."""
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn


class Pipeline(nn.Module):
    def __init__(self, model: nn.Module, preprocess: Optional[nn.Module] = None, postprocess: Optional[nn.Module] = None):
        super().__init__()
        from .postprocess import NoOpPostProcess
        self.preprocess = preprocess if preprocess is not None else nn.Identity()
        self.model = model
        # no post-process given -> the pass-through module, as the reference (predictor/base_module.py:45-46):
        # forward() and detect() then share one path
        self.postprocess = postprocess if postprocess is not None else NoOpPostProcess()
        # the column map of the result rows travels with the network (reference models/detection/yolov3.py:56-60)
        self.output_format = dict(getattr(model, "output_format", {}))
        self.task = getattr(model, "task", "detection")
        self.eval()

    @torch.no_grad()
    def forward(self, input: torch.Tensor, *args, **kwargs):
        return self.postprocess(self.model(self.preprocess(input)), *args, **kwargs)

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


_REQUIRED_KEYS = {"detection": ("bounding_box", "class_confidence"), "classification": ("class_label", "class_confidence")}


class BasePredictor(Pipeline):
    """Pipeline under the reference's constructor contract (predictor/base_module.py:37-102):
      * the network must carry `task` ("detection" | "classification", anything else is a RuntimeError) and
        `output_format` (AssertionError when missing) holding the keys the task requires (RuntimeError otherwise);
      * a pre-process other than nn.Identity and a post-process other than NoOpPostProcess must have an annotated
        `forward` (RuntimeError); the pre-process annotations must read input: Tensor -> Tensor (TypeError);
      * `postprocess=None` means NoOpPostProcess, `preprocess=None` means nn.Identity;
      * forward() runs under the caller's grad mode (Pipeline wraps it in no_grad; the reference does not)."""

    def __init__(self, model: nn.Module, preprocess: Optional[nn.Module] = None, postprocess: Optional[nn.Module] = None):
        from .postprocess import NoOpPostProcess
        if not hasattr(model, "task"):
            raise AttributeError("model definition '%s' has no 'task' attribute" % type(model).__name__)
        assert hasattr(model, "output_format"), \
            "model definition '%s' doesn't have 'output_format' attribute" % type(model)
        super().__init__(model, preprocess, postprocess)
        self.task = model.task
        if type(self.postprocess) is not NoOpPostProcess and not getattr(self.postprocess.forward, "__annotations__", None):
            raise RuntimeError("please annotate postprocess with %s" % ["input", "return"])
        if type(self.preprocess) is not nn.Identity:
            ann = getattr(self.preprocess.forward, "__annotations__", None)
            if not ann:
                raise RuntimeError("please annotate preprocess with %s" % ["input", "return"])
            for key in ("input", "return"):
                if key not in ann:
                    raise TypeError("missing preprocess annotation(s) : %s" % key)
                if ann[key] != torch.Tensor:
                    raise TypeError("type mismatch for `%s`, expects %s got %s" % (key, torch.Tensor, ann[key]))
        if self.task not in _REQUIRED_KEYS:
            raise RuntimeError("unknown task of '%s'" % self.task)
        for key in _REQUIRED_KEYS[self.task]:
            if key not in self.output_format:
                raise RuntimeError("expects output_format for task %s to have `%s` key, got output_format: %s"
                                   % (self.task, key, self.output_format))

    def forward(self, input: torch.Tensor, *args, **kwargs):
        return self.postprocess(self.model(self.preprocess(input)), *args, **kwargs)


def create_predictor(model_components, **kwargs) -> BasePredictor:
    """reference predictor/base_module.py:105-113: components -> BasePredictor (with its checks)."""
    def pick(name):
        if isinstance(model_components, dict):
            return model_components.get(name)
        return getattr(model_components, name, None)
    return BasePredictor(pick("network"), preprocess=pick("preprocess"), postprocess=pick("postprocess"))


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
