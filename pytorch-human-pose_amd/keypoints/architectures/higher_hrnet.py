"""HigherHRNet with the reference's module interface, executed by the gfx950 HIP engine.

Stands in for `/root/reference/src/keypoints/architectures/higher_hrnet.py:47-81`
(`HigherHRNet(num_kpts, C).forward(images) -> ([hm_1/4, hm_1/2], tags_1/4)`); it registers
in `KeypointsConfig.architectures` (keypoints/config.py:93-95) under the same name, owns
parameters under the reference's 1810 state-dict keys, and loads `higher_hrnet_32.pt`
unchanged.  Inference runs entirely in csrc/libhhrnet.so (bf16 MFMA kernels, BN folded);
there is no ATen / CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import Tensor, nn

from ... import _lib
from .spec import attach_modules, higher_hrnet_rows


class _NetHandle:
    """Owns the hh_net* so the nn.Module needs no __del__ of its own."""

    def __init__(self, lib, ptr):
        self.lib = lib
        self.ptr = ptr
        if not self.ptr:
            raise _lib.HHError(lib.hh_last_error().decode())

    def __del__(self):
        ptr, self.ptr = self.ptr, None
        if ptr:
            self.lib.hh_destroy(ptr)


class EngineModule(nn.Module):
    """nn.Module whose parameters live under the reference's state-dict keys and whose forward runs in the HIP engine."""

    def _init_engine(self, rows, create) -> None:
        attach_modules(self, rows)
        self._lib = _lib.load()
        self._handle = _NetHandle(self._lib, create(self._lib))
        self._h = self._handle.ptr
        self._dirty = True
        self.use_graph = True
        self.register_load_state_dict_post_hook(lambda m, _k: m.mark_dirty())

    # ---- engine plumbing
    def mark_dirty(self) -> None:
        """Call after mutating parameters in place (optimizer steps, init_weights)."""
        self._dirty = True

    def _apply(self, fn, *a, **kw):
        self._dirty = True
        return super()._apply(fn, *a, **kw)

    def train(self, mode: bool = True):
        self._dirty = True
        return super().train(mode)

    def engine_param_names(self) -> list[str]:
        n = self._lib.hh_num_params(self._h)
        return [self._lib.hh_param_name(self._h, i).decode() for i in range(n)]

    def sync_weights(self) -> None:
        """state_dict -> hh_load_weights (fp32 host copies) -> hh_finalize (BN fold + bf16 pack)."""
        for name, t in self.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(self._lib.hh_load_weights(self._h, name.encode(), a.ctypes.data, shape, a.ndim))
        _lib.check(self._lib.hh_finalize(self._h))
        self._dirty = False

    def forward_flops(self, B: int, H: int, W: int) -> float:
        return float(self._lib.hh_forward_flops(self._h, B, H, W))

    def workspace_bytes(self) -> int:
        return int(self._lib.hh_workspace_bytes(self._h))

    def _check_input(self, images: Tensor) -> Tensor:
        if self.training:
            raise NotImplementedError(
                f"{type(self).__name__}: the fused inference engine folds BatchNorm and keeps no gradients; in .train() mode call "
                "the module (forward), which runs keypoints/train_net.py, or .eval() first"
            )
        if not images.is_cuda:
            raise _lib.HHError(f"{type(self).__name__} forward needs a CUDA/HIP tensor: there is no CPU path")
        if self._dirty:
            self.sync_weights()
        x = images.contiguous().float()
        assert x.dim() == 4 and x.shape[1] == 3
        return x

    # ---- debug taps for the parity tests
    def set_taps(self, enable: bool) -> None:
        self._lib.hh_set_taps(self._h, int(enable))

    def read_taps(self) -> dict[str, np.ndarray]:
        out = {}
        for i in range(self._lib.hh_num_taps(self._h)):
            shape = (C.c_int64 * 4)()
            _lib.check(self._lib.hh_tap_shape(self._h, i, shape))
            a = np.empty(tuple(shape), np.float32)
            _lib.check(self._lib.hh_tap_read(self._h, i, a.ctypes.data))
            out[self._lib.hh_tap_name(self._h, i).decode()] = a
        return out


class HigherHRNet(EngineModule):
    """`dtype="bf16"` (default) or `"fp8"` (BASELINE.json configs[4]: OCP e4m3 MFMA operands and activations; an extension, the
    reference has one precision).  An fp8 net needs `calibrate(images)` once after its weights are loaded."""

    DTYPES = {"bf16": 1, "fp8": 2}

    def __init__(self, num_kpts: int, C: int = 32, dtype: str = "bf16"):
        super().__init__()
        self.num_kpts = num_kpts
        self.C = C
        self.num_deconv_layers = 1
        self.engine_dtype = dtype
        code = self.DTYPES[dtype]
        self._init_engine(higher_hrnet_rows(num_kpts, C), lambda lib: lib.hh_create(num_kpts, C, code))

    def calibrate(self, images: Tensor, rounds: int = 2) -> None:
        """fp8 only: per-tensor activation scales from forwards over `images` [B,3,H,W] (hh_calibrate).  Weight changes
        (load_state_dict, optimizer steps) invalidate it."""
        x = self._check_input(images)
        B, _, H, W = x.shape
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hh_calibrate(self._h, x.data_ptr(), B, H, W, rounds, stream))

    def forward_raw(self, images: Tensor, out: tuple[Tensor, Tensor] | None = None) -> tuple[Tensor, Tensor]:
        """-> (init_heatmaps [B,2K,H/4,W/4], deconv_heatmaps [B,K,H/2,W/2]) fp32.
        `out` = preallocated result tensors (keeps the pointers, hence the cached hipGraph, stable)."""
        x = self._check_input(images)
        B, _, H, W = x.shape
        K = self.num_kpts
        if out is not None:
            init, dec = out
            assert init.shape == (B, 2 * K, H // 4, W // 4) and dec.shape == (B, K, H // 2, W // 2)
            assert init.is_contiguous() and dec.is_contiguous() and init.dtype == dec.dtype == torch.float32
        else:
            init = torch.empty((B, 2 * K, H // 4, W // 4), device=x.device, dtype=torch.float32)
            dec = torch.empty((B, K, H // 2, W // 2), device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hh_forward(self._h, x.data_ptr(), B, H, W, init.data_ptr(), dec.data_ptr(),
                                            int(self.use_graph), stream))
        return init, dec

    def forward(self, images: Tensor) -> tuple[list[Tensor], Tensor]:
        if self.training:  # batch-stat BatchNorm, differentiable: keypoints/train_net.py on the training kernels
            from ..train_net import higher_hrnet_train_forward
            if not images.is_cuda:
                raise _lib.HHError("HigherHRNet forward needs a CUDA/HIP tensor: there is no CPU path")
            self._dirty = True  # parameters / running statistics change under training: re-fold before the next eval forward
            return higher_hrnet_train_forward(self, images)
        init, dec = self.forward_raw(images)
        K = self.num_kpts
        return [init[:, :K], dec[:, :K]], init[:, K:]
