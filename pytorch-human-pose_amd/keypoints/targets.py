"""Training targets of the associative-embedding loss (SURVEY.md §8 a20 inputs), host side.

Restated from the text of `src/keypoints/datasets/coco.py:76-137,140-164` (that module needs albumentations and
pycocotools and cannot be imported here, so these two small generators are checked by property tests only).
"""
from __future__ import annotations

import numpy as np
import torch


class HeatmapGenerator:
    """coco.py:76-121: max-composited Gaussian blobs (sigma 2, support 6*sigma+3) per visible in-bounds joint."""

    def __init__(self, num_kpts: int, size: int, sigma: float = 2):
        self.num_kpts, self.size = num_kpts, size
        self.h = self.w = size
        if sigma < 0:
            sigma = size / 64
        self.sigma = sigma
        x = np.arange(0, 6 * sigma + 3, 1, float)
        y = x[:, np.newaxis]
        x0 = y0 = 3 * sigma + 1
        self.gauss = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma**2))

    def __call__(self, joints: np.ndarray) -> np.ndarray:
        hms = np.zeros((self.num_kpts, self.h, self.w), dtype=np.float32)
        s = self.sigma
        for person in joints:
            for k in range(self.num_kpts):
                x, y, vis = person[k]
                if vis <= 0 or x < 0 or y < 0 or x >= self.w or y >= self.h:
                    continue
                xmin, ymin = int(np.round(x - 3 * s - 1)), int(np.round(y - 3 * s - 1))
                xmax, ymax = int(np.round(x + 3 * s + 2)), int(np.round(y + 3 * s + 2))
                c, d = max(0, -xmin), min(xmax, self.w) - xmin
                a, b = max(0, -ymin), min(ymax, self.h) - ymin
                cc, dd = max(0, xmin), min(xmax, self.w)
                aa, bb = max(0, ymin), min(ymax, self.h)
                hms[k, aa:bb, cc:dd] = np.maximum(hms[k, aa:bb, cc:dd], self.gauss[a:b, c:d])
        return hms


class JointsGenerator:
    """coco.py:124-137: integer (x, y, 1) for visible in-bounds joints, (0, 0, 0) otherwise; people without any
    visible joint are dropped (`joints.sum(axis=(1, 2)) > 0`)."""

    def __init__(self, size: int = 512):
        self.h = self.w = size

    def __call__(self, joints: np.ndarray) -> np.ndarray:
        joints = np.array(joints, dtype=np.float64, copy=True)
        for i in range(len(joints)):
            for k, pt in enumerate(joints[i]):
                x, y, vis = int(pt[0]), int(pt[1]), pt[2]
                joints[i, k] = (x, y, 1) if (vis > 0 and 0 <= x < self.w and 0 <= y < self.h) else (0, 0, 0)
        return joints[joints.sum(axis=(1, 2)) > 0].astype(np.int32)


def collate(samples):
    """coco.py:140-164: [(image, [hm per stage], [mask per stage], [joints per stage])] -> batch tuple."""
    n = len(samples[0][1])
    images = torch.from_numpy(np.stack([s[0] for s in samples]))
    heatmaps = [torch.from_numpy(np.stack([s[1][i] for s in samples])) for i in range(n)]
    masks = [torch.from_numpy(np.stack([s[2][i] for s in samples])) for i in range(n)]
    joints = [[s[3][i] for s in samples] for i in range(n)]
    return images, heatmaps, masks, joints
