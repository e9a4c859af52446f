"""Drop-in for `src/keypoints/loss.py`: same classes and call signatures, HIP kernels underneath.

Every loss value is a 0-dim tensor that takes part in torch autograd: the kernels compute the value and the gradient
in one pass (`hh_loss_heatmaps`, `hh_loss_ae_grouping`), `backward` only scales the stored gradient by the incoming
one (GradScaler's factor arrives that way).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import Tensor
from torch.nn.modules.loss import _Loss

from .. import _lib


def _scratch(device, n: int) -> Tensor:
    return torch.empty(max(1024, n), device=device, dtype=torch.float64)


def _plane_view(t: Tensor, what: str) -> Tensor:
    """[B,K,h,w] fp32 with contiguous planes; only the batch stride may differ (channel slices stay views)."""
    if not t.is_cuda:
        raise _lib.HHError(f"{what} must be a CUDA/HIP tensor: there is no CPU path")
    _, _, h, w = t.shape
    if t.dtype != torch.float32 or t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w or t.stride(0) % 4:
        t = t.float().contiguous()
    return t


class _HeatmapsLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, mask: Tensor) -> Tensor:
        lib = _lib.load()
        p = _plane_view(pred.detach(), "pred_heatmaps")
        B, K, h, w = p.shape
        t = target.to(p.device, torch.float32).contiguous()
        m = mask.to(p.device, torch.float32).contiguous()
        if t.shape != p.shape or m.shape != (B, h, w):
            raise ValueError(f"HeatmapsLoss: shapes {tuple(pred.shape)}, {tuple(target.shape)}, {tuple(mask.shape)} do not match")
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        grad = torch.empty((B, K, h, w), device=p.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        stream = torch.cuda.current_stream(p.device).cuda_stream
        with torch.cuda.device(p.device):
            _lib.check(lib.hh_loss_heatmaps(p.data_ptr(), p.stride(0), t.data_ptr(), m.data_ptr(), B, K, h, w, loss.data_ptr(),
                                            grad.data_ptr() if grad is not None else None, K * h * w,
                                            _scratch(p.device, 0).data_ptr(), stream))
        ctx.grad, ctx.dtype = grad, pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        return (ctx.grad * g).to(ctx.dtype), None, None


class HeatmapsLoss(_Loss):
    """loss.py:6-16"""

    def forward(self, pred_heatmaps: Tensor, target_heatmaps: Tensor, mask: Tensor) -> Tensor:
        return _HeatmapsLossFn.apply(pred_heatmaps, target_heatmaps, mask)


def pack_joints(joints: list, K: int, h: int, w: int) -> tuple[np.ndarray, np.ndarray]:
    """list over images of int [P_b,K,3] (x, y, vis) -> (int32 [B,Pmax,K,3] zero padded, int32 [B] counts).
    A visible joint outside the map raises IndexError like `pred_tags[i, k, y, x]` would (loss.py:30); negative
    coordinates follow python indexing."""
    B = len(joints)
    counts = np.array([len(j) for j in joints], np.int32)
    packed = np.zeros((B, max(1, int(counts.max()) if B else 1), K, 3), np.int32)
    for b, j in enumerate(joints):
        if len(j) == 0:
            continue
        j = np.asarray(j.cpu().numpy() if torch.is_tensor(j) else j)
        if j.shape[1:] != (K, 3):
            raise ValueError(f"joints[{b}] has shape {j.shape}, expected [P,{K},3]")
        j = j.astype(np.int64)
        vis = j[..., 2] > 0
        x, y = np.where(j[..., 0] < 0, j[..., 0] + w, j[..., 0]), np.where(j[..., 1] < 0, j[..., 1] + h, j[..., 1])
        if np.any(vis & ((x < 0) | (x >= w) | (y < 0) | (y >= h))):
            raise IndexError(f"joints[{b}]: a visible joint lies outside the {h}x{w} tag map")
        packed[b, : len(j), :, 0], packed[b, : len(j), :, 1], packed[b, : len(j), :, 2] = x, y, vis
    return packed, counts


class _AEGroupingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_tags: Tensor, packed: Tensor, counts: Tensor) -> tuple[Tensor, Tensor]:
        lib = _lib.load()
        t = _plane_view(pred_tags.detach(), "pred_tags")
        B, K, h, w = t.shape
        P = packed.shape[1]
        out = torch.empty(2, device=t.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        need = ctx.needs_input_grad[0]
        # push and pull get separate gradient maps: the caller may weight the two losses differently
        gpush = torch.zeros((B, K, h, w), device=t.device, dtype=torch.float32) if need else None
        gpull = torch.zeros((B, K, h, w), device=t.device, dtype=torch.float32) if need else None
        with torch.cuda.device(t.device):
            sc = _scratch(t.device, 2 * B)
            if need:
                _lib.check(lib.hh_loss_ae_grouping(t.data_ptr(), t.stride(0), packed.data_ptr(), counts.data_ptr(), B, P, K, h, w,
                                                   out.data_ptr(), gpush.data_ptr(), K * h * w, 1.0, 0.0, sc.data_ptr(), stream))
                _lib.check(lib.hh_loss_ae_grouping(t.data_ptr(), t.stride(0), packed.data_ptr(), counts.data_ptr(), B, P, K, h, w,
                                                   out.data_ptr(), gpull.data_ptr(), K * h * w, 0.0, 1.0, sc.data_ptr(), stream))
            else:
                _lib.check(lib.hh_loss_ae_grouping(t.data_ptr(), t.stride(0), packed.data_ptr(), counts.data_ptr(), B, P, K, h, w,
                                                   out.data_ptr(), None, 0, 0.0, 0.0, sc.data_ptr(), stream))
        ctx.gpush, ctx.gpull, ctx.dtype = gpush, gpull, pred_tags.dtype
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_push: Tensor, g_pull: Tensor):
        return (ctx.gpush * g_push + ctx.gpull * g_pull).to(ctx.dtype), None, None


class DeviceJoints:
    """The joints of a batch already packed and uploaded (`upload_joints`): lets a caller keep the host -> device copy out
    of the step, e.g. to capture the whole training step in a hipGraph.  Not in the reference (it passes host lists)."""

    def __init__(self, packed: Tensor, counts: Tensor):
        self.packed, self.counts = packed, counts

    def __len__(self) -> int:
        return int(self.counts.shape[0])


def upload_joints(joints: list, K: int, h: int, w: int, device) -> DeviceJoints:
    packed, counts = pack_joints(joints, K, h, w)
    return DeviceJoints(torch.from_numpy(packed).to(device), torch.from_numpy(counts).to(device))


class AEGroupingLoss(_Loss):
    """loss.py:19-61 -> (push_loss / batch, pull_loss / batch)"""

    def forward(self, pred_tags: Tensor, joints) -> tuple[Tensor, Tensor]:
        if not pred_tags.is_cuda:
            raise _lib.HHError("pred_tags must be a CUDA/HIP tensor: there is no CPU path")
        B, K, h, w = pred_tags.shape
        if len(joints) != B:
            raise ValueError(f"joints has {len(joints)} entries for a batch of {B}")
        if not isinstance(joints, DeviceJoints):
            joints = upload_joints(joints, K, h, w, pred_tags.device)
        return _AEGroupingFn.apply(pred_tags, joints.packed, joints.counts)


class AEKeypointsLoss(_Loss):
    """loss.py:64-93"""

    def __init__(self) -> None:
        super().__init__()
        self.heatmaps_losses = torch.nn.ModuleList([HeatmapsLoss() for _ in range(2)])
        self.tags_loss = AEGroupingLoss()

    def calculate_loss(self, stages_pred_kpts_heatmaps: list[Tensor], pred_tags_heatmaps: Tensor,
                       stages_target_heatmaps: list[Tensor], masks: list[Tensor], joints: list):
        heatmap_losses = [self.heatmaps_losses[i](stages_pred_kpts_heatmaps[i], stages_target_heatmaps[i], masks[i])
                          for i in range(len(stages_target_heatmaps))]
        push_loss, pull_loss = self.tags_loss(pred_tags_heatmaps, joints[0])
        return heatmap_losses, [push_loss * 1e-3], [pull_loss * 1e-3]
