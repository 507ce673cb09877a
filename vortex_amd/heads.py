"""Drop-in for YoloV3Layer of the reference (networks/modules/heads/detection/yolov3.py:192-304)
and yolo_feature_maps (networks/modules/utils/darknet.py:136-143).  The eval branch — grid/anchor
decode of one scale — runs in the HIP kernel vx_yolo_head_decode; the conv blocks of the head stay
in PyTorch-ROCm (out of scope)."""
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from . import ops


def yolo_feature_maps(input_size: int, backbone_stages: Tuple[int, int, int] = (3, 4, 5)) -> List[Tuple[int, int]]:
    assert len(backbone_stages) == 3
    s3, s4, s5 = (2 ** x for x in backbone_stages)
    return [(input_size // s3, input_size // s3), (input_size // s4, input_size // s4),
            (input_size // s5, input_size // s5)]


class YoloV3Layer(nn.Module):
    def __init__(self, img_size: Union[int, List[int]], grids: Tuple[int, int], mask: List[int],
                 anchors: List[Tuple[int, int]], n_classes: int):
        super(YoloV3Layer, self).__init__()
        if type(img_size) == list:
            img_size = max(img_size)
        stride = img_size // grids[0]
        self.all_anchors_grid = [(w / stride, h / stride) for w, h in anchors]
        ref_anchors = torch.zeros(len(self.all_anchors_grid), 4)
        ref_anchors[:, 2:] = torch.tensor(self.all_anchors_grid)
        self.register_buffer("ref_anchors", ref_anchors)
        self.anchor_mask = mask
        anchors = torch.Tensor([anchors[i] for i in mask])
        self.register_buffer("anchors", anchors)
        self.n_anchors = len(mask)
        self.n_classes = n_classes
        nx, ny = grids
        self.nx, self.ny = nx, ny
        stride_t = torch.tensor([img_size / max(grids)])                       # :236
        yv, xv = torch.meshgrid([torch.arange(ny).float(), torch.arange(nx).float()], indexing="ij")
        self.register_buffer("grid_xy", torch.stack((xv, yv), 2).view((1, 1, ny, nx, 2)))
        self.register_buffer("anchor_wh", (anchors / stride_t).view(1, self.n_anchors, 1, 1, 2))
        self.register_buffer("stride", stride_t)
        self.register_buffer("anchor_vec", self.anchors / self.stride)
        self.n_predict = self.n_classes + 5
        # host copies: kernel arguments
        self._anchor_wh_host = (anchors / stride_t).reshape(-1).tolist()
        self._stride_host = float(stride_t.item())

    def forward(self, prediction: torch.Tensor, out: torch.Tensor = None, row_offset: int = 0):
        assert len(prediction.shape) == 4
        batch_size, ch, nx, ny = prediction.shape
        if self.training:
            # training branch (no grid decode) is not on the inference path: plain torch, as the reference
            p = prediction.view(batch_size, self.n_anchors, self.n_predict, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            xy = torch.sigmoid(p[..., 0:2])
            conf = torch.sigmoid(p[..., 4]).view(batch_size, self.n_anchors, nx, ny, 1)
            return torch.cat((torch.cat((xy, p[..., 2:4]), 4), conf, torch.sigmoid(p[..., 5:])), -1)
        if self.n_anchors != 3 or nx != ny or ch != 3 * self.n_predict:
            raise RuntimeError("vx_yolo_head_decode expects 3 anchors and a square grid, got %s" % (tuple(prediction.shape),))
        return ops.yolo_head_decode(prediction, self._anchor_wh_host, self._stride_host, out=out, row_offset=row_offset)


class YoloV3HeadDecode(nn.Module):
    """The eval tail of the reference's YoloV3 model in one kernel launch: its three YoloV3Layer decodes and
    `torch.cat((h1, h2, h3), 1)` (networks/models/detection/yolov3.py:72-76; layers and masks as built at
    heads/detection/yolov3.py:319-343).  forward(h1, h2, h3) takes the raw conv outputs coarse -> fine and returns
    f32[B, R, 5+C], the tensor YoloV3PostProcess consumes -- the same bits as three YoloV3Layer calls."""

    def __init__(self, img_size: int, n_classes: int, anchors: List[Tuple[int, int]] = None):
        super().__init__()
        self.img_size, self.n_classes = int(img_size), int(n_classes)
        self.anchors = ops.COCO_ANCHORS if anchors is None else [tuple(a) for a in anchors]
        assert len(self.anchors) == 9

    def forward(self, h1: torch.Tensor, h2: torch.Tensor, h3: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if h1.size(1) != 3 * (self.n_classes + 5):
            raise RuntimeError("vx_yolo_head_decode3 expects 3 anchors x (5 + %d) channels, got %s"
                               % (self.n_classes, tuple(h1.shape),))
        return ops.yolo_head_decode3((h1, h2, h3), self.img_size, self.anchors, out=out)
