"""ctypes front-end of oracle/decode_oracle.c (the CPU restatement of the reference decode).

TEST INFRASTRUCTURE ONLY -- see the header of decode_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "decode_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_munkres.restype = C.c_int
        _LIB.orc_match_by_tag.restype = C.c_int
        _LIB.orc_parse.restype = C.c_int
    return _LIB


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def bilinear(x: np.ndarray, H: int, W: int) -> np.ndarray:
    """x [C,h,w] fp32 -> [C,H,W]; torch CPU bilinear(align_corners=False) bit-for-bit."""
    x = _f32(x)
    c, h, w = x.shape
    out = np.empty((c, H, W), np.float32)
    lib().orc_bilinear(_p(x, C.c_float), c, h, w, _p(out, C.c_float), H, W, H * W, 1)
    return out


def aggregate(hm_q, hm_h, tags_list):
    """results.py:225-234 -> (hm_full [K,H,W], tags_full [K,H,W,E])."""
    hm_q, hm_h = _f32(hm_q), _f32(hm_h)
    tags_list = [_f32(t) for t in tags_list]
    K, hq, wq = hm_q.shape
    E = len(tags_list)
    H, W = 4 * hq, 4 * wq
    hm_full = np.empty((K, H, W), np.float32)
    tags_full = np.empty((K, H, W, E), np.float32)
    arr = (C.POINTER(C.c_float) * E)(*[_p(t, C.c_float) for t in tags_list])
    lib().orc_aggregate(_p(hm_q, C.c_float), _p(hm_h, C.c_float), arr, E, K, hq, wq, _p(hm_full, C.c_float),
                        _p(tags_full, C.c_float))
    return hm_full, tags_full


def munkres(cost: np.ndarray):
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    r, c = cost.shape
    pairs = np.zeros((max(r, c), 2), np.int32)
    n = lib().orc_munkres(_p(cost, C.c_double), r, c, _p(pairs, C.c_int32))
    if n < 0:
        raise RuntimeError("munkres did not terminate")
    return pairs[:n]


def top_k(hm_full, tags_full, max_people: int):
    hm_full, tags_full = _f32(hm_full), _f32(tags_full)
    K, H, W = hm_full.shape
    E = tags_full.shape[-1]
    nms = np.empty_like(hm_full)
    lib().orc_nms(_p(hm_full, C.c_float), K, H, W, _p(nms, C.c_float))
    tags_k = np.empty((K, max_people, E), np.float32)
    coords_k = np.empty((K, max_people, 2), np.int32)
    scores_k = np.empty((K, max_people), np.float32)
    lib().orc_topk(_p(nms, C.c_float), _p(tags_full, C.c_float), K, H, W, E, max_people, _p(tags_k, C.c_float),
                   _p(coords_k, C.c_int32), _p(scores_k, C.c_float))
    return tags_k, coords_k, scores_k


def match_by_tag(tags_k, coords_k, scores_k, det_thr: float, tag_thr: float):
    tags_k, scores_k = _f32(tags_k), _f32(scores_k)
    coords_k = np.ascontiguousarray(coords_k, dtype=np.int32)
    K, maxp, E = tags_k.shape
    out = np.zeros((maxp, K, 3 + E), np.float32)
    P = lib().orc_match_by_tag(_p(tags_k, C.c_float), _p(coords_k, C.c_int32), _p(scores_k, C.c_float), K, maxp, E,
                               C.c_double(det_thr), C.c_double(tag_thr), _p(out, C.c_float))
    return out[:P]


def adjust(grouped, hm_full):
    grouped = _f32(grouped).copy()
    hm_full = _f32(hm_full)
    P, K, D = grouped.shape
    lib().orc_adjust(_p(grouped, C.c_float), P, K, D - 3, _p(hm_full, C.c_float), hm_full.shape[1], hm_full.shape[2])
    return grouped


def parse(hm_full, tags_full, max_people=30, det_thr=0.1, tag_thr=1.0, adjust=True, refine=True, return_topk=False):
    """MPPEHeatmapParser(...).parse (grouping.py:252-283) on full-resolution maps."""
    hm_full, tags_full = _f32(hm_full), _f32(tags_full)
    K, H, W = hm_full.shape
    E = tags_full.shape[-1]
    joints = np.zeros((max(max_people, 1), K, 3 + E), np.float32)
    scores = np.zeros((max(max_people, 1),), np.float32)
    tags_k = np.empty((K, max_people, E), np.float32)
    coords_k = np.empty((K, max_people, 2), np.int32)
    scores_k = np.empty((K, max_people), np.float32)
    P = lib().orc_parse(_p(hm_full, C.c_float), _p(tags_full, C.c_float), K, H, W, E, max_people, C.c_double(det_thr),
                        C.c_double(tag_thr), int(adjust), int(refine), _p(joints, C.c_float), _p(scores, C.c_float),
                        _p(tags_k, C.c_float), _p(coords_k, C.c_int32), _p(scores_k, C.c_float))
    if P < 0:  # no group: grouping.py:262-269 builds the pseudo-person by concatenating int32 with float32 -> float64 arrays,
        joints = joints[:1].astype(np.float64)  # writes the Python float 0.01 as its score and averages that in float64
        joints[..., 2] = 0.01
        scores, P = joints[..., 2].mean(1), 1
    if return_topk:
        return joints[:P], scores[:P], (tags_k, coords_k, scores_k)
    return joints[:P], scores[:P]


def decode(hm_q, hm_h, tags_list, **kw):
    """from_preds' aggregation + parse (results.py:225-238) from raw network outputs."""
    full, tfull = aggregate(hm_q, hm_h, tags_list)
    return parse(full, tfull, **kw)


def transform_coords(xy, center, scale, hm_size):
    """results.py:158-171 with get_affine_transform(inverse=True), rot=0: oracle/transforms.py (cv2.getAffineTransform's LU solve
    restated; PARITY UNPINNED: no cv2).  [..., 2] float32 in, float32 [-1, 2] out like the reference's in-place float32 rows."""
    from . import transforms
    return transforms.transform_coords(_f32(xy).reshape(-1, 2), center, scale, hm_size)


def get_multi_scale_size(h: int, w: int, input_size: int, current_scale: float, min_scale: float):
    """base/transforms/utils.py:60-86 restated: the short image side maps to input_size (rounded up
    to a multiple of 64 at min_scale), the long side to the next multiple of 64, both then scaled
    by current_scale/min_scale.  Returns ((w_resized, h_resized), center, (scale_w, scale_h))."""
    base = int((min_scale * input_size + 63) // 64 * 64)
    ratio_num, ratio_den = current_scale, min_scale
    portrait = w < h
    short, long_ = (w, h) if portrait else (h, w)
    short_r = int(base * ratio_num / ratio_den)
    long_r = int(int((base / short * long_ + 63) // 64 * 64) * ratio_num / ratio_den)
    long_scale = long_r / short_r * short
    center = (int(w / 2.0 + 0.5), int(h / 2.0 + 0.5))
    if portrait:
        return (short_r, long_r), center, (short, long_scale)
    return (long_r, short_r), center, (long_scale, short)
