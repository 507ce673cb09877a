"""Prior-box (FPN-SSD / RetinaFace) decoder and post-process.  The reference's own modules
(networks/modules/postprocess/retinaface.py, fpn_ssd) are absent from the snapshot; this follows the
build-authored spec of SURVEY.md §8a-9 (see oracle/vortex_oracle.py:prior_box_decode) and plugs into
the same BatchedNMSPostProcess protocol (base_postprocess.py:110-138)."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .postprocess import BatchedNMSPostProcess, _f


class PriorBoxDecoder(nn.Module):
    def __init__(self, priors: torch.Tensor, n_landmarks: int = 0, variances=(0.1, 0.2), threshold: bool = True):
        super().__init__()
        self.register_buffer("priors", priors.float().contiguous())
        self.n_landmarks = n_landmarks
        self.variances = variances
        self.threshold = threshold

    def decode_batch(self, input: torch.Tensor, score_threshold) -> ops.Candidates:
        if input.dim() != 3:
            raise RuntimeError("this routine expects predictions is a 3-dimensional tensor! got %s dimension" % input.dim())
        return ops.prior_decode(input, self.priors, self.n_landmarks, _f(score_threshold), self.threshold, self.variances)

    def forward(self, input: torch.Tensor, score_threshold: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        assert input.size(0) == 1                      # scripts/export_check/README.md:55
        c = self.decode_batch(input, score_threshold)
        meta = c.meta.cpu()
        ops.raise_on_status(int(meta[-1]))
        k = int(meta[0])
        return c.boxes[0, :k], c.scores[0, :k], c.cls[0, :k], c.det[:, :k]


class PriorBoxPostProcess(BatchedNMSPostProcess):
    def __init__(self, priors: torch.Tensor, n_landmarks: int = 0, variances=(0.1, 0.2), threshold: bool = True,
                 nms: bool = True, top_k: Optional[int] = None):
        super().__init__(decoder=PriorBoxDecoder(priors, n_landmarks, variances, threshold), nms=nms, top_k=top_k)

    def n_classes_hint(self, input) -> int:
        return int(input.size(-1)) - 4 - 2 * self.decoder.n_landmarks - 1
