"""COCO result packing of the evaluation harness (`src/keypoints/bin/eval.py:18-49`).

Only the packing and the image-sharded loop live here: AP itself is computed by pycocotools 2.0.7 in the reference
(`bin/eval.py:52-65`), a third-party evaluator that is not in this image, so nothing below restates it.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .distributed import gather_results, shard_range


def image_id_from_path(image_filepath: str) -> int:
    """bin/eval.py:23: COCO file stems are zero-padded image ids."""
    return int(Path(image_filepath).stem.lstrip("0"))


def pack_coco_results(image_id: int, kpts_coords: np.ndarray, obj_scores: np.ndarray) -> list[dict]:
    """bin/eval.py:27-48: one entry per person, keypoints = [x, y, 1] * K (float64), score = the person's score
    (`scores.mean()` of a scalar in the reference, i.e. the score itself, as a python float)."""
    results = []
    for i in range(len(obj_scores)):
        kpts = kpts_coords[i]
        coco_kpts = np.zeros((len(kpts) * 3,))
        coco_kpts[::3] = kpts[:, 0]
        coco_kpts[1::3] = kpts[:, 1]
        coco_kpts[2::3] = 1
        results.append({"image_id": int(image_id), "category_id": 1, "keypoints": coco_kpts.tolist(),
                        "score": np.asarray(obj_scores[i]).mean().item()})
    return results


def evaluate_images(model, images, image_ids, rank: int = 0, world_size: int = 1, multi_scale=None, batch: int = 32) -> list[dict] | None:
    """evaluate_dataset (bin/eval.py:18-49) over in-memory images, sharded by image across ranks (§8e: contiguous
    slices, no data-path collective; the packed lists are gathered to rank 0, other ranks get None).
    `multi_scale` = tuple of scales -> `model.call_multi_scale` (the cfg-4 extension) instead of `model(...)`.
    `batch` > 1 routes the single-scale case through `model.infer_images` (identical per-image results, one forward and one
    decode per shape bucket); `batch` = 1 is the reference's image-by-image loop."""
    local = []
    mine = list(shard_range(len(images), rank, world_size))
    if multi_scale or batch <= 1:
        for idx in mine:
            res = model.call_multi_scale(images[idx], None, multi_scale) if multi_scale else model(images[idx], None)
            local += pack_coco_results(image_ids[idx], res.kpts_coords, res.obj_scores)
    else:  # batched behind the same per-image results: shape buckets of up to `batch` images per forward + decode
        for idx, res in zip(mine, model.infer_images([images[i] for i in mine], max_batch=batch)):
            local += pack_coco_results(image_ids[idx], res.kpts_coords, res.obj_scores)
    return gather_results(local)
