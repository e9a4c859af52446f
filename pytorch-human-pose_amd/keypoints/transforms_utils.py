"""Inference resize geometry and coordinate un-warp (host side, numpy).

Restates `/root/reference/src/base/transforms/utils.py:5-97`.  `cv2` is not available in
the build/run images, so `warp_affine` is this repo's own bilinear warp with cv2's
conventions (inverse map, zero border) but float weights instead of cv2's 5-bit fixed
point tables: image preprocessing is not bit-identical to opencv 4.9.
"""
from __future__ import annotations

import numpy as np

COCO_FLIP_INDEX = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]  # keypoints/transforms.py:11
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)  # keypoints/model.py:49
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def get_multi_scale_size(image: np.ndarray, input_size: int, current_scale: float, min_scale: float):
    """utils.py:60-86 -> ((w_resized, h_resized), center, (scale_w, scale_h))"""
    h, w = image.shape[:2]
    base = int((min_scale * input_size + 63) // 64 * 64)
    portrait = w < h
    short, long_ = (w, h) if portrait else (h, w)
    short_r = int(base * current_scale / min_scale)
    long_r = int(int((base / short * long_ + 63) // 64 * 64) * current_scale / min_scale)
    long_scale = long_r / short_r * short
    center = (int(w / 2.0 + 0.5), int(h / 2.0 + 0.5))
    if portrait:
        return (short_r, long_r), center, (short, long_scale)
    return (long_r, short_r), center, (long_scale, short)


def affine_matrix(center, scale, output_size, inverse: bool = False) -> np.ndarray:
    """get_affine_transform(center, scale, rot=0, output_size) (utils.py:25-57): with rot = 0 the
    three point pairs define an isotropic scale r = dst_w / scale_w about center <-> (dst_w/2, dst_h/2)."""
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    r = dst_w / float(scale[0])
    if inverse:
        r = 1.0 / r
        return np.array([[r, 0.0, center[0] - r * dst_w * 0.5], [0.0, r, center[1] - r * dst_h * 0.5]], np.float64)
    return np.array([[r, 0.0, dst_w * 0.5 - r * center[0]], [0.0, r, dst_h * 0.5 - r * center[1]]], np.float64)


def warp_affine(image: np.ndarray, m: np.ndarray, size) -> np.ndarray:
    """cv2.warpAffine(image, m, size) semantics (m maps src->dst, bilinear, constant-0 border)."""
    w_out, h_out = int(size[0]), int(size[1])
    a = np.vstack([m, [0, 0, 1]])
    inv = np.linalg.inv(a)
    xs, ys = np.meshgrid(np.arange(w_out, dtype=np.float64), np.arange(h_out, dtype=np.float64))
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0)[..., None]
    fy = (sy - y0)[..., None]
    h, w = image.shape[:2]
    img = image.astype(np.float32)
    if img.ndim == 2:
        img = img[..., None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(image.dtype)


def resize_align_multi_scale(image: np.ndarray, input_size: int, current_scale: float, min_scale: float):
    """utils.py:89-97"""
    size, center, scale = get_multi_scale_size(image, input_size, current_scale, min_scale)
    return warp_affine(image, affine_matrix(center, scale, size), size), center, scale
