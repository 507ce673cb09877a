"""Host-only checks of the caller contract (reference predictor/base_module.py:37-113): BasePredictor's defaults,
required attributes and annotation checks; no kernel is launched."""
import pytest
import torch
import torch.nn as nn

import vortex_amd as V
from vortex_amd.predictor import Pipeline

FMT = {"bounding_box": {"indices": [0, 1, 2, 3], "axis": 1}, "class_confidence": {"indices": [4], "axis": 1},
       "class_label": {"indices": [5], "axis": 1}}


class Net(nn.Module):
    task = "detection"
    output_format = FMT

    def forward(self, x):
        return x * 2


def test_create_predictor_without_postprocess_is_a_pass_through():
    p = V.create_predictor(dict(network=Net()))
    assert isinstance(p, V.BasePredictor) and isinstance(p.postprocess, V.NoOpPostProcess)
    assert isinstance(p.preprocess, nn.Identity) and p.task == "detection" and not p.training
    x = torch.arange(12.0).reshape(2, 1, 6)             # two images, one row of six columns each
    assert torch.equal(p(x), x * 2)
    res = p.detect(x)                                   # forward() and detect() share the pass-through path
    assert len(res) == 2 and res[0]["bounding_box"].tolist() == [[0.0, 2.0, 4.0, 6.0]]


def test_base_predictor_runs_under_the_callers_grad_mode():
    w = torch.ones(3, requires_grad=True)

    class G(Net):
        def forward(self, x):
            return x * w
    assert V.BasePredictor(G())(torch.ones(3)).requires_grad          # the reference does not wrap forward in no_grad
    assert not Pipeline(G())(torch.ones(3)).requires_grad


def test_base_predictor_contract_errors():
    class NoTask(nn.Module):
        output_format = FMT
    with pytest.raises(AttributeError):
        V.BasePredictor(NoTask())

    class NoFormat(nn.Module):
        task = "detection"
    with pytest.raises(AssertionError):
        V.BasePredictor(NoFormat())

    class Odd(Net):
        task = "segmentation"
    with pytest.raises(RuntimeError, match="unknown task"):
        V.BasePredictor(Odd())

    class Missing(Net):
        output_format = {"bounding_box": FMT["bounding_box"]}
    with pytest.raises(RuntimeError, match="class_confidence"):
        V.BasePredictor(Missing())

    class Cls(Net):
        task = "classification"
        output_format = {"class_label": {"indices": [0], "axis": 0}, "class_confidence": {"indices": [1], "axis": 0}}
    V.BasePredictor(Cls())

    class Bare(nn.Module):
        def forward(self, x):
            return x
    with pytest.raises(RuntimeError, match="annotate"):
        V.BasePredictor(Net(), postprocess=Bare())
    with pytest.raises(RuntimeError, match="annotate"):
        V.BasePredictor(Net(), preprocess=Bare())

    class Half(nn.Module):
        def forward(self, input: torch.Tensor):
            return input
    with pytest.raises(TypeError, match="missing"):
        V.BasePredictor(Net(), preprocess=Half())

    class Wrong(nn.Module):
        def forward(self, input: int) -> torch.Tensor:
            return input
    with pytest.raises(TypeError, match="mismatch"):
        V.BasePredictor(Net(), preprocess=Wrong())
    # the shipped modules satisfy the checks
    V.BasePredictor(Net(), preprocess=V.Normalize(), postprocess=V.YoloV3PostProcess(64))
