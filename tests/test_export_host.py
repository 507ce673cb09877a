"""CPU tests of the exporter-compatibility layer (vortex_amd/export.py): the custom ops exist with the expected
schemas, shape propagation works on fake tensors (what torch.export / torch.compile need), and a real CPU tensor is
refused loudly (no CPU fallback)."""
import pytest
import torch

import vortex_amd as V
from vortex_amd import export  # noqa: F401  (registers the ops)


def test_ops_registered_with_schemas():
    for name in ("nms", "batched_nms", "normalize"):
        assert hasattr(torch.ops.vortex_amd, name)
    s = str(torch.ops.vortex_amd.nms.default._schema)
    assert "Tensor boxes" in s and "float iou_threshold" in s and "-> Tensor" in s
    s = str(torch.ops.vortex_amd.normalize.default._schema)
    assert "float[] mean" in s and "bool flip" in s


def test_fake_tensor_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.symbolic_shapes import ShapeEnv
    with FakeTensorMode(shape_env=ShapeEnv(), allow_non_fake_inputs=False) as mode:
        img = mode.from_tensor(torch.zeros(2, 8, 12, 3, dtype=torch.uint8))
        y = torch.ops.vortex_amd.normalize(img, [0.5] * 3, [0.5] * 3, 255, False)
        assert tuple(y.shape) == (2, 3, 8, 12) and y.dtype == torch.float32
        boxes, scores = mode.from_tensor(torch.zeros(7, 4)), mode.from_tensor(torch.zeros(7))
        keep = torch.ops.vortex_amd.nms(boxes, scores, 0.5)
        assert keep.dtype == torch.int64 and keep.dim() == 1
        keep = torch.ops.vortex_amd.batched_nms(boxes, scores, mode.from_tensor(torch.zeros(7, dtype=torch.int64)), 0.5)
        assert keep.dtype == torch.int64 and keep.dim() == 1


def test_cpu_tensors_refused():
    with pytest.raises(V._lib.VortexHipError):
        torch.ops.vortex_amd.nms(torch.rand(5, 4), torch.rand(5), 0.5)
    with pytest.raises(V._lib.VortexHipError):
        torch.ops.vortex_amd.normalize(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), [0.5] * 3, [0.5] * 3, 255, False)


def test_use_custom_ops_hook():
    post = V.YoloV3PostProcess(img_size=64)
    export.use_custom_ops(post)
    assert post.nms.nms_fn is torch.ops.vortex_amd.batched_nms
    assert isinstance(export.register_onnx_symbolics(), bool)
