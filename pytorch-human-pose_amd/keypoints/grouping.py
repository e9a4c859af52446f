"""Associative-embedding decode on the GPU behind the reference's parser interface.

Stands in for `/root/reference/src/keypoints/grouping.py:62-283` (`MPPEHeatmapParser`): same
constructor, same `parse(kpts_hms[K,H,W], tags_hms[K,H,W,E], adjust, refine)` contract and
return types `(np.float32[P,K,3+E], np.float32[P])`.  NMS / top-k / Hungarian tag matching /
adjust / refine all run in csrc/libhhrnet.so (hh_parse / hh_decode); nothing is computed
on the host and the full-resolution maps are never copied to it (the reference moves
36-54 MB per image device->host, grouping.py:271-272).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import Tensor

from .. import _lib


class _DecHandle:
    def __init__(self, lib, num_kpts, max_people, det_thr, tag_thr):
        self.lib = lib
        self.ptr = lib.hh_decoder_create(num_kpts, max_people, float(det_thr), float(tag_thr))
        if not self.ptr:
            raise _lib.HHError(lib.hh_last_error().decode())

    def __del__(self):
        ptr, self.ptr = self.ptr, None
        if ptr:
            self.lib.hh_decoder_destroy(ptr)


class MPPEHeatmapParser:
    joints_order: list[int] = [0, 1, 2, 3, 4, 5, 6, 11, 12, 7, 8, 9, 10, 13, 14, 15, 16]  # grouping.py:63-65

    def __init__(self, num_kpts: int, max_num_people: int = 30, det_thr: float = 0.1, tag_thr: float = 1.0):
        self.num_kpts = num_kpts
        self.max_num_people = max_num_people
        self.det_thr = det_thr
        self.tag_thr = tag_thr
        self._lib = _lib.load()
        self._handle = _DecHandle(self._lib, num_kpts, max_num_people, det_thr, tag_thr)
        self._h = self._handle.ptr

    def set_exact_topk(self, enable: bool) -> None:
        """hh_decoder_set_exact_topk: True = every NMS tile is processed, so `last_top_k` / `top_k` return the reference's full
        candidate lists; False (default) = tiles that cannot reach det_thr are skipped (same parse results, faster)."""
        self._lib.hh_decoder_set_exact_topk(self._h, int(enable))

    # ------------------------------------------------------------------ device-side batch API
    def _outputs(self, B: int, E: int, device):
        K, M = self.num_kpts, self.max_num_people
        joints = torch.empty((B, M, K, 3 + E), device=device, dtype=torch.float32)
        scores = torch.empty((B, M), device=device, dtype=torch.float32)
        num = torch.empty((B,), device=device, dtype=torch.int32)
        flags = torch.empty((B,), device=device, dtype=torch.int32)  # HH_DECODE_* bits per image
        return joints, scores, num, flags

    def parse_batch_device(self, kpts_hms: Tensor, tags_hms: Tensor, adjust: bool = True, refine: bool = True):
        """kpts_hms [B,K,H,W], tags_hms [B,K,H,W,E] (full resolution) -> device (joints, scores, num_people)."""
        if not kpts_hms.is_cuda:
            raise _lib.HHError("MPPEHeatmapParser needs CUDA/HIP tensors: there is no CPU path")
        hm = kpts_hms.contiguous().float()
        tg = tags_hms.contiguous().float()
        B, K, H, W = hm.shape
        E = tg.shape[-1]
        assert K == self.num_kpts and tuple(tg.shape[:4]) == (B, K, H, W)
        joints, scores, num, flags = self._outputs(B, E, hm.device)
        stream = torch.cuda.current_stream(hm.device).cuda_stream
        with torch.cuda.device(hm.device):
            _lib.check(self._lib.hh_parse(self._h, hm.data_ptr(), tg.data_ptr(), E, B, H, W, int(adjust), int(refine),
                                          joints.data_ptr(), scores.data_ptr(), num.data_ptr(), flags.data_ptr(), stream))
        return joints, scores, num, flags

    def decode_batch_device(self, hm_q: Tensor, hm_h: Tensor, tags_q: list[Tensor], adjust: bool = True, refine: bool = True):
        """Fused results.py:225-238 + parse from raw net outputs:
        hm_q [B,K,h,w], hm_h [B,K,2h,2w], tags_q: E tensors [B,K,h,w] (channel-slice views are fine)."""
        B, K, hq, wq = hm_q.shape
        E = len(tags_q)

        def view(t, h, w):  # [B,K,h,w] with contiguous planes; only the batch stride may differ
            if t.dtype != torch.float32 or t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w:
                t = t.contiguous().float()
            return t

        hm_q, hm_h = view(hm_q, hq, wq), view(hm_h, 2 * hq, 2 * wq)
        tags_q = [view(t, hq, wq) for t in tags_q]
        if not hm_q.is_cuda:
            raise _lib.HHError("MPPEHeatmapParser needs CUDA/HIP tensors: there is no CPU path")
        joints, scores, num, flags = self._outputs(B, E, hm_q.device)
        tptr = (C.c_void_p * E)(*[t.data_ptr() for t in tags_q])
        tbs = (C.c_int64 * E)(*[t.stride(0) for t in tags_q])
        stream = torch.cuda.current_stream(hm_q.device).cuda_stream
        self._keep = (hm_q, hm_h, tags_q)
        with torch.cuda.device(hm_q.device):
            _lib.check(self._lib.hh_decode(self._h, hm_q.data_ptr(), hm_q.stride(0), hm_h.data_ptr(), hm_h.stride(0), tptr, tbs,
                                           E, B, hq, wq, int(adjust), int(refine), joints.data_ptr(), scores.data_ptr(),
                                           num.data_ptr(), flags.data_ptr(), stream))
        return joints, scores, num, flags

    @staticmethod
    def to_lists(joints: Tensor, scores: Tensor, num: Tensor, flags: Tensor):
        """Device results -> the reference's per-image `(joints [P,K,3+E], scores [P])` pairs.  Raises if the assignment solver
        gave up on an image.  An image without any group comes back as the reference returns it (grouping.py:262-269): ONE
        pseudo-person in float64 arrays (np.concatenate of int32 coordinates with float32 scores / tags promotes), its joint
        scores the Python float 0.01 and the person score their float64 mean."""
        j, s, n, f = joints.cpu().numpy(), scores.cpu().numpy(), num.cpu().numpy(), flags.cpu().numpy()
        if (f & 2).any():
            raise _lib.HHError(f"decode: the assignment solver hit its iteration guard on image(s) {np.nonzero(f & 2)[0].tolist()}")
        out = []
        for b in range(j.shape[0]):
            jb, sb = j[b, : n[b]].copy(), s[b, : n[b]].copy()
            if f[b] & 1:
                jb = jb.astype(np.float64)
                jb[..., 2] = 0.01
                sb = jb[..., 2].mean(1)
            out.append((jb, sb))
        return out

    def last_top_k(self, B: int, E: int):
        """tags_k [B,K,M,E], coords_k [B,K,M,2], scores_k [B,K,M] of the last call (top_k, grouping.py:147-170)."""
        K, M = self.num_kpts, self.max_num_people
        tags_k = np.empty((B, K, M, E), np.float32)
        coords_k = np.empty((B, K, M, 2), np.int32)
        scores_k = np.empty((B, K, M), np.float32)
        _lib.check(self._lib.hh_decoder_read_topk(self._h, tags_k.ctypes.data, coords_k.ctypes.data, scores_k.ctypes.data))
        return tags_k, coords_k, scores_k

    # ------------------------------------------------------------------ the reference's interface
    def parse(self, kpts_hms: Tensor, tags_hms: Tensor, adjust: bool = True, refine: bool = True):
        """grouping.py:252-283"""
        if tags_hms.dim() == 3:
            tags_hms = tags_hms[..., None]
        out = self.parse_batch_device(kpts_hms[None], tags_hms[None], adjust, refine)
        return self.to_lists(*out)[0]

    def top_k(self, kpts_hms: Tensor, tags_hms: Tensor):
        """grouping.py:147-170 -> (tags_k [K,M,E], coords_k [K,M,2] int32 (x,y), scores_k [K,M])"""
        if tags_hms.dim() == 3:
            tags_hms = tags_hms[..., None]
        self.set_exact_topk(True)
        try:
            self.parse_batch_device(kpts_hms[None], tags_hms[None], False, False)
            t, c, s = self.last_top_k(1, tags_hms.shape[-1])
        finally:
            self.set_exact_topk(False)
        return t[0], c[0], s[0]
