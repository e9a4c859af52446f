"""Inference resize geometry and coordinate un-warp (host side, numpy).

Mirrors `/root/reference/src/base/transforms/utils.py:5-97`.  The OpenCV arithmetic behind it (cv2.getAffineTransform's LU
solve, cv2.warpAffine's matrix inversion and fixed-point bilinear) runs behind the C-ABI (hh_get_affine_transform,
hh_invert_affine, hh_warp_affine_u8 / hh_preprocess_u8); `cv2` itself is in neither the build nor the run image, so parity with
opencv 4.9 is UNPINNED -- the tests compare with the independent restatement in oracle/transforms.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib

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
    """get_affine_transform(center, scale, rot=0, output_size, inverse) (utils.py:25-57) -> the float64 2x3 matrix
    cv2.getAffineTransform returns for the reference's float32 point pairs (hh_get_affine_transform: OpenCV's 6x6 LU solve;
    for rot = 0 it is the isotropic scale r = dst_w / scale_w about center <-> (dst_w/2, dst_h/2) up to the solve's rounding)."""
    m = np.empty(6, np.float64)
    _lib.check(_lib.load().hh_get_affine_transform(float(center[0]), float(center[1]), float(scale[0]), float(output_size[0]),
                                                   float(output_size[1]), int(inverse), m.ctypes.data_as(C.POINTER(C.c_double))))
    return m.reshape(2, 3)


def dst_to_src_matrix(center, scale, size) -> np.ndarray:
    """What cv2.warpAffine(image, get_affine_transform(...), size) does first: the forward matrix inverted in float64 with
    OpenCV's own formula (hh_invert_affine) -> contiguous float64 [2,3], the `dst_to_src` argument of the preprocessing kernels."""
    fwd = np.ascontiguousarray(affine_matrix(center, scale, size).reshape(6))
    inv = np.empty(6, np.float64)
    dp = C.POINTER(C.c_double)
    _lib.check(_lib.load().hh_invert_affine(fwd.ctypes.data_as(dp), inv.ctypes.data_as(dp)))
    return inv.reshape(2, 3)


def warp_affine(image: np.ndarray, m: np.ndarray, size, device="cuda:0") -> np.ndarray:
    """cv2.warpAffine(image, m, size) for uint8 HWC images (m maps src->dst; bilinear, constant-0 border) on the GPU
    (hh_warp_affine_u8: OpenCV's fixed-point arithmetic).  There is no host fallback."""
    import torch
    w_out, h_out = int(size[0]), int(size[1])
    dp = C.POINTER(C.c_double)
    fwd = np.ascontiguousarray(np.asarray(m, np.float64).reshape(6))
    inv = np.empty(6, np.float64)
    lib = _lib.load()
    _lib.check(lib.hh_invert_affine(fwd.ctypes.data_as(dp), inv.ctypes.data_as(dp)))
    img = np.ascontiguousarray(image, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("warp_affine: uint8 [h, w, 3] images only")
    raw = torch.from_numpy(img).to(device)
    out = torch.empty((h_out, w_out, 3), device=raw.device, dtype=torch.uint8)
    with torch.cuda.device(raw.device):
        _lib.check(lib.hh_warp_affine_u8(raw.data_ptr(), img.shape[0], img.shape[1], inv.ctypes.data_as(dp), out.data_ptr(), h_out, w_out,
                                         torch.cuda.current_stream(raw.device).cuda_stream))
    return out.cpu().numpy()


def resize_align_multi_scale(image: np.ndarray, input_size: int, current_scale: float, min_scale: float, device="cuda:0"):
    """utils.py:89-97"""
    size, center, scale = get_multi_scale_size(image, input_size, current_scale, min_scale)
    return warp_affine(image, affine_matrix(center, scale, size), size, device), center, scale
