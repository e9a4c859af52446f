"""ClassificationHRNet with the reference's module interface, executed by the gfx950 HIP engine.

Stands in for `/root/reference/src/classification/architectures/hrnet.py:64-74`
(`ClassificationHRNet(C, num_classes).forward(images) -> logits`), BASELINE.json configs[0].
Same state-dict keys as the reference (backbone with a 4-scale last fusion + classification_head.*).
"""
from __future__ import annotations

import torch
from torch import Tensor

from ... import _lib
from ...keypoints.architectures.higher_hrnet import EngineModule
from ...keypoints.architectures.spec import classification_hrnet_rows


class ClassificationHRNet(EngineModule):
    def __init__(self, C: int = 32, num_classes: int = 1000):
        super().__init__()
        self.C = C
        self.num_classes = num_classes
        self.stages_C = [C, 2 * C, 4 * C, 8 * C]
        self._init_engine(classification_hrnet_rows(C, num_classes), lambda lib: lib.hh_create_classifier(C, num_classes, 1))

    def forward(self, images: Tensor) -> Tensor:
        x = self._check_input(images)
        B, _, H, W = x.shape
        logits = torch.empty((B, self.num_classes), device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hh_forward_classifier(self._h, x.data_ptr(), B, H, W, logits.data_ptr(), stream))
        return logits
