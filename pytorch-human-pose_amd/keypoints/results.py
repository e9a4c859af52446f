"""Result container of one inference call.

Stands in for the non-plotting part of `/root/reference/src/keypoints/results.py:175-263`
(`InferenceKeypointsResult.from_preds`): stage-heatmap aggregation and decode run fused on
the GPU (hh_decode), the coordinate un-warp (results.py:158-171,189-201) on the host.
Plotting / OKS helpers are out of scope.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch
from torch import Tensor

from .. import _lib
from .grouping import MPPEHeatmapParser


def transform_coords(kpts_coords: np.ndarray, center, scale, output_size) -> np.ndarray:
    """results.py:158-171 on an [..., 2] array (inverse of the resize affine, float64 like cv2)."""
    xy = np.ascontiguousarray(kpts_coords, dtype=np.float32).reshape(-1, 2)
    out = np.empty((xy.shape[0], 2), np.float64)
    _lib.check(_lib.load().hh_transform_coords(xy.ctypes.data, xy.shape[0], float(center[0]), float(center[1]), float(scale[0]),
                                               float(output_size[0]), float(output_size[1]), out.ctypes.data))
    return out.reshape(kpts_coords.shape).astype(kpts_coords.dtype)


class KeypointsResult:
    """Validation-time result of one image (`results.py:70-124`, without the plotting helpers): the stage heatmaps and tags of
    the net as they come out of `forward` (batch dim kept: [1,K,h,w]), decoded on demand by `set_preds()` with this result's
    own thresholds (validation uses max_num_people=20, det_thr=0.1, tag_thr=1.0, `keypoints/module.py:100-108`).  Coordinates
    are model-input pixels (no un-warp: the validation image IS the model input)."""

    def __init__(self, model_input_image: Tensor, kpts_heatmaps: list[Tensor], tags_heatmaps: Tensor, limbs, max_num_people: int = 30,
                 det_thr: float = 0.05, tag_thr: float = 0.5, parser: MPPEHeatmapParser | None = None):
        self.model_input_image = model_input_image
        self._kpts_heatmaps = kpts_heatmaps
        self._tags_heatmaps = tags_heatmaps
        self.num_kpts = kpts_heatmaps[0].shape[1]
        self.limbs = limbs
        self.max_num_people, self.det_thr, self.tag_thr = max_num_people, det_thr, tag_thr
        self.hm_parser = parser or MPPEHeatmapParser(self.num_kpts, max_num_people, det_thr, tag_thr)

    def _assign(self, joints: np.ndarray, scores: np.ndarray) -> None:
        self.kpts_coords, self.kpts_scores, self.kpts_tags, self.obj_scores = joints[..., :2], joints[..., 2], joints[..., 3:], scores

    def set_preds(self) -> None:
        """results.py:94-124: stage average, resize, parse -- fused in hh_decode."""
        out = self.hm_parser.decode_batch_device(self._kpts_heatmaps[0], self._kpts_heatmaps[1], [self._tags_heatmaps], adjust=True, refine=True)
        self._assign(*self.hm_parser.to_lists(*out)[0])

    @staticmethod
    def set_preds_batch(results: list["KeypointsResult"], stages: list[Tensor], tags: Tensor) -> None:
        """The same for every image of a batch in ONE hh_decode call (identical per-image results: images are independent)."""
        if not results:
            return
        parser = results[0].hm_parser
        for r, (j, s) in zip(results, parser.to_lists(*parser.decode_batch_device(stages[0], stages[1], [tags], adjust=True, refine=True))):
            r._assign(j, s)


@dataclass
class InferenceKeypointsResult:
    raw_image: np.ndarray | None
    annot: list | None
    model_input_image: Tensor | None
    kpts_coords: np.ndarray  # [P,K,2] raw-image pixels
    kpts_scores: np.ndarray  # [P,K]
    kpts_tags: np.ndarray  # [P,K,E]
    obj_scores: np.ndarray  # [P]
    det_thr: float
    tag_thr: float
    limbs: list = field(default_factory=list)
    _stage_hms: list | None = None
    _tags: list | None = None

    @classmethod
    def from_preds(cls, raw_image, annot, model_input_image: Tensor, kpts_heatmaps: list[Tensor], tags_heatmaps: list[Tensor],
                   limbs, scale, center, det_thr: float = 0.05, tag_thr: float = 0.5, max_num_people: int = 30,
                   parser: MPPEHeatmapParser | None = None) -> "InferenceKeypointsResult":
        """results.py:203-263 for batch size 1 (the reference's inference wrapper is single-image)."""
        K = tags_heatmaps[0].shape[1]
        parser = parser or MPPEHeatmapParser(K, max_num_people, det_thr, tag_thr)
        img_h, img_w = model_input_image.shape[-2:]
        out = parser.decode_batch_device(kpts_heatmaps[0], kpts_heatmaps[1], tags_heatmaps, adjust=True, refine=True)
        joints, scores = parser.to_lists(*out)[0]
        coords = transform_coords(joints[..., :2], center, scale, (img_w, img_h))
        return cls(raw_image, annot, model_input_image, coords, joints[..., 2], joints[..., 3:], scores, det_thr, tag_thr, limbs,
                   kpts_heatmaps, tags_heatmaps)

    # visualisation-only views of the reference (resized maps on the host); not on the hot path
    @property
    def kpts_heatmaps(self) -> np.ndarray:
        f = torch.nn.functional.interpolate
        h, w = self.model_input_image.shape[-2:]
        a, b = self._stage_hms
        avg = (f(a, size=list(b.shape[-2:]), mode="bilinear", align_corners=False) + b) / 2
        return f(avg, size=[h, w], mode="bilinear", align_corners=False)[0].cpu().numpy()

    @property
    def tags_heatmaps(self) -> np.ndarray:
        f = torch.nn.functional.interpolate
        h, w = self.model_input_image.shape[-2:]
        return f(self._tags[0], size=[h, w], mode="bilinear", align_corners=False)[0].cpu().numpy()
