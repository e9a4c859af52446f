"""Training targets of the associative-embedding loss (SURVEY.md §8 a20 inputs), host side.

What the reference's dataset code produces for one image (`src/keypoints/datasets/coco.py:76-137,140-164`; that module
needs albumentations and pycocotools and cannot be imported here, so these generators carry property tests only):
  * per joint type a heatmap that is the pixel-wise MAXIMUM over people of an isotropic Gaussian bump
    exp(-(dx^2 + dy^2) / (2 sigma^2)) truncated to the square |dx|, |dy| <= 3 sigma + 1 around the joint,
  * per person the integer pixel of every visible joint for the grouping loss.
Written here from that definition (offset grids and window clipping), synthetic-batch plumbing for bench.py --train.
"""
from __future__ import annotations

import numpy as np
import torch


class HeatmapGenerator:
    """Max-composite of truncated Gaussian bumps; `sigma < 0` means size / 64 like the reference's default."""

    def __init__(self, num_kpts: int, size: int, sigma: float = 2):
        self.num_kpts, self.size = num_kpts, size
        self.sigma = size / 64 if sigma < 0 else sigma
        # bump sampled on the integer grid 0 .. 6 sigma + 2 with its centre at 3 sigma + 1 (float64, like np.exp on a float grid)
        self.reach = 3 * self.sigma + 1
        grid = np.arange(0, 6 * self.sigma + 3, 1, float)
        d2 = (grid[None, :] - self.reach) ** 2 + (grid[:, None] - self.reach) ** 2
        self.bump = np.exp(-d2 / (2 * self.sigma ** 2))

    def _window(self, centre: float):
        """Clip the bump's support along one axis: (first map index, one past the last, first bump index)."""
        start = int(np.round(centre - self.reach))
        stop = int(np.round(centre + self.reach + 1))
        lo, hi = max(start, 0), min(stop, self.size)
        return lo, hi, lo - start

    def __call__(self, joints: np.ndarray) -> np.ndarray:
        maps = np.zeros((self.num_kpts, self.size, self.size), dtype=np.float32)
        for person in np.asarray(joints):
            for k, (x, y, vis) in enumerate(person[: self.num_kpts]):
                if vis <= 0 or not (0 <= x < self.size and 0 <= y < self.size):
                    continue
                x_lo, x_hi, bx = self._window(x)
                y_lo, y_hi, by = self._window(y)
                view = maps[k, y_lo:y_hi, x_lo:x_hi]
                np.maximum(view, self.bump[by:by + (y_hi - y_lo), bx:bx + (x_hi - x_lo)], out=view)
        return maps


class JointsGenerator:
    """Integer (x, y, 1) per visible joint inside the map, (0, 0, 0) otherwise; people with no such joint are dropped."""

    def __init__(self, size: int = 512):
        self.size = size

    def __call__(self, joints: np.ndarray) -> np.ndarray:
        j = np.asarray(joints, dtype=np.float64)
        xy = np.trunc(j[..., :2])  # int() of the reference truncates toward zero
        inside = (j[..., 2] > 0) & (xy >= 0).all(-1) & (xy < self.size).all(-1)
        out = np.zeros(j.shape[:2] + (3,), np.int32)
        out[..., :2] = np.where(inside[..., None], xy, 0).astype(np.int32)
        out[..., 2] = inside
        return out[inside.any(-1)]


def collate(samples):
    """[(image, [hm per stage], [mask per stage], [joints per stage])] -> (images, [hm], [mask], [[joints per image]])."""
    stages = range(len(samples[0][1]))
    stack = lambda pick: torch.from_numpy(np.stack([pick(s) for s in samples]))  # noqa: E731
    return (stack(lambda s: s[0]), [stack(lambda s, i=i: s[1][i]) for i in stages], [stack(lambda s, i=i: s[2][i]) for i in stages],
            [[s[3][i] for s in samples] for i in stages])
