"""Callers of the hot path, kept API-compatible with the reference:
  predictor/base_module.py:9-113   BasePredictor (preprocess -> model -> postprocess), create_predictor
  predictor/utils.py:11-33         get_prediction_results (np.take per output_format key)
The model (backbone + head convs) is any torch nn.Module running on PyTorch-ROCm."""
from collections import OrderedDict
from typing import Dict, List, Type, Union

import numpy as np
import torch
import torch.nn as nn

from .postprocess import NoOpPostProcess


class BasePredictor(nn.Module):
    postprocess_signature = {'input': torch.Tensor, 'return': Union[torch.Tensor, List[torch.Tensor]]}
    preprocess_signature = {'input': torch.Tensor, 'return': torch.Tensor}

    def __init__(self, model: Type[nn.Module], preprocess: Union[Type[nn.Module]] = None,
                 postprocess: Union[Type[nn.Module]] = None):
        super(BasePredictor, self).__init__()
        self.model = model
        if preprocess is None:
            preprocess = nn.Identity()
        if postprocess is None:
            postprocess = NoOpPostProcess()
        self.preprocess = preprocess
        self.postprocess = postprocess
        self.task = model.task
        if type(self.postprocess) != NoOpPostProcess:
            if not len(self.postprocess.forward.__annotations__):
                raise RuntimeError("please annotate postprocess with %s" % BasePredictor.postprocess_signature.keys())
        if type(self.preprocess) != nn.modules.linear.Identity:
            if not len(self.preprocess.forward.__annotations__):
                raise RuntimeError("please annotate postprocess with %s" % BasePredictor.preprocess_signature.keys())
            annotations = self.preprocess.forward.__annotations__
            for key, value in BasePredictor.preprocess_signature.items():
                if key not in annotations.keys():
                    raise TypeError("missing preprocess annotation(s) : %s" % key)
                if value != annotations[key]:
                    raise TypeError("type mismatch for `%s`, expects %s got %s" % (key, value, annotations[key]))
        assert hasattr(self.model, 'output_format'), f"model definition '{type(model)}' doesn't have 'output_format' attribute"
        self.output_format = dict(self.model.output_format)
        if self.task == "detection":
            required = ["bounding_box", "class_confidence"]
        elif self.task == "classification":
            required = ["class_label", "class_confidence"]
        else:
            raise RuntimeError("unknown task of '%s'" % self.task)
        for elem in required:
            if elem not in self.output_format:
                raise RuntimeError("expects output_format for task %s to have `%s` key, got output_format: %s" % (
                    self.task, elem, self.output_format))
        self.eval()

    def forward(self, input: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        x = self.preprocess(input)
        x = self.model(x)
        x = self.postprocess(x, *args, **kwargs)
        return x


def create_predictor(model_components, **kwargs):
    get = (lambda k: model_components[k] if k in model_components else None) if isinstance(model_components, dict) \
        else (lambda k: getattr(model_components, k, None))
    return BasePredictor(get("network"), preprocess=get("preprocess"), postprocess=get("postprocess"))


def get_prediction_results(results, output_format: Dict[str, dict]) -> List[Dict[str, np.ndarray]]:
    fmt = {}
    for k, v in output_format.items():
        ind = v["indices"]
        if isinstance(ind, dict):
            ind = [x for x in range(ind['start'], ind['end'])]
        fmt[k] = (ind, v["axis"])
    output_fields = sorted(fmt.keys())
    final_results = []
    for result in results:
        if isinstance(result, torch.Tensor):
            result = result.detach().cpu().numpy()
        temp = OrderedDict()
        for key in output_fields:
            temp[key] = np.take(result, indices=fmt[key][0], axis=fmt[key][1]) if all(result.shape) else None
        final_results.append(temp)
    return final_results
