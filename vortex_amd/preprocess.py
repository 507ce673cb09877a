"""Drop-in for the reference's networks/modules/preprocess/normalizer.py (Normalize :63-77,
FlipNormalize :79-95, get_preprocess :102-121) and preprocess/__init__.py:47-55 (names
`normalizer`, `normalize`, `flip_normalizer`, `flip_normalize`).  The arithmetic runs in the
HIP kernel vx_preprocess_u8hwc_f32nchw; same constructor arguments, buffers, annotations and
assertions as the reference."""
import torch
import torch.nn as nn

from . import ops


class Normalize(nn.Module):
    """uint8 NHWC -> float NCHW, ((x / scaler) - mean) / std.  Returns a contiguous NCHW tensor
    (the reference returns an NCHW-shaped view of NHWC memory; values are identical)."""
    __constants__ = ["mean", "std", "scaler"]

    def __init__(self, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], scaler=255):
        super(Normalize, self).__init__()
        self.register_buffer("mean", torch.as_tensor(mean, dtype=torch.float))
        self.register_buffer("std", torch.as_tensor(std, dtype=torch.float))
        self.register_buffer("scaler", torch.as_tensor(scaler, dtype=torch.int))
        self._flip = False
        self._host = None

    def _consts(self):
        # host copies of the (tiny) buffers, read once: they are kernel arguments
        if self._host is None:
            self._host = (self.mean.detach().cpu().tolist(), self.std.detach().cpu().tolist(),
                          int(self.scaler.item()))
        return self._host

    def _apply(self, fn, *a, **k):
        self._host = None
        return super()._apply(fn, *a, **k)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # reference to_tensor asserts (normalizer.py:23-26)
        assert len(input.size()) == 4, "to_tensor only support 4-dimension tensor (NHWC) as input, "\
            "got size of %s" % list(input.size())
        assert input.size(3) == 3 or input.size(3) == 1, "to_tensor only support NHWC input layout, "\
            "got size of %s" % list(input.size())
        if self._flip:
            assert input.size(3) == 3, "only supports 3 color channels with HWC format"
        if input.size(3) != 3:
            raise NotImplementedError("libvortex_hip pre-process handles 3-channel images")
        mean, std, scaler = self._consts()
        return ops.preprocess(input, mean, std, scaler, flip=self._flip)


class FlipNormalize(Normalize):
    """Channel reversal (RGB<->BGR) in front of Normalize (normalizer.py:79-95)."""

    def __init__(self, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], scaler=255, flip_dim=-1):
        super(FlipNormalize, self).__init__(mean, std, scaler)
        if flip_dim not in (-1, 3):
            raise NotImplementedError("only the channel dimension can be flipped")
        self.flip_dim = flip_dim
        self._flip = True

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return super().forward(input)


preproc_map = dict(
    normalizer=Normalize, normalize=Normalize,
    flip_normalizer=FlipNormalize, flip_normalize=FlipNormalize,
)


def get_preprocess(preprocess: str, **kwargs):
    assert preprocess in preproc_map, "unsupported preproc f{preprocess}"
    normalizer = preproc_map[preprocess]
    normalization_args = kwargs["input_normalization"] if "input_normalization" in kwargs else kwargs
    return normalizer(**normalization_args)
