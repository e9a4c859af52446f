"""Train-mode forward of HigherHRNet on the HIP building blocks.

The layer graph is the reference's (`src/keypoints/architectures/hrnet.py:29-385`, `higher_hrnet.py:7-81`), walked over
the shim's parameter tree.  Convolutions (forward, data gradient, weight gradient) and train-mode BatchNorm (+ residual,
+ ReLU, forward and backward) are `torch.autograd.Function`s over the C-ABI ops of `train_ops`; torch autograd is only the
tape, and the glue between kernels (nearest upsample, sums of the fusion layers, channel concat / slicing, bias adds) are
torch elementwise ops on the same bf16 channels_last tensors.  Activations are bf16 (the reference trains under fp16
autocast, `module.py:50`), parameters and their gradients fp32, so torch optimizers, GradScaler-free bf16 training and
DistributedDataParallel (gradient all-reduce over RCCL) work on the module unchanged.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import train_ops as ops


class _ResBox:
    """The gradient a residual unit's skip connection carries (hrnet.py:62-74,108-124: `out += identity`), handed from the last
    BatchNorm's backward straight to the first conv's: that conv's data-gradient launch starts its accumulators from it (the
    residual input of the conv kernels), so dL/dx = conv1's data gradient + skip gradient is rounded to bf16 once and autograd's
    separate elementwise add per unit (111 launches per step) is gone.  The unit's backward always runs bn_last -> ... -> conv1."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


def _box_put(box, g) -> None:
    """The last BatchNorm's backward leaves the unit's skip gradient for the first conv's backward; a gradient still waiting there belongs
    to a backward whose first conv never ran (a partial torch.autograd.grad) and must not be added to this one."""
    if box.g is not None:
        raise RuntimeError("train_net: a stale skip gradient is still waiting for its unit's first convolution (a previous backward did not "
                           "reach it); HH_TRAIN_NO_RESBOX=1 routes the skip gradient through autograd instead")
    box.g = g


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, stride: int, pad, packed_fwd=None, packed_bwd=None, box=None):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.pad, ctx.packed_bwd, ctx.box = stride, pad, packed_bwd, box
        return ops.conv2d(x, w, stride, pad=pad, packed=packed_fwd)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        skip = None
        if ctx.box is not None:  # the unit's skip gradient (left there by its last BatchNorm's backward, which has run)
            if ctx.box.g is None:  # e.g. torch.autograd.grad towards inputs the last BatchNorm does not reach: the skip path would be dropped silently
                raise RuntimeError("train_net: the residual unit's skip gradient is missing (its last BatchNorm's backward has not run); "
                                   "HH_TRAIN_NO_RESBOX=1 routes it through autograd instead")
            skip, ctx.box.g = ctx.box.g, None
        dx = ops.conv2d(dy, w, ctx.stride, data_grad=True, pad=ctx.pad, packed=ctx.packed_bwd, res=skip) if ctx.needs_input_grad[0] else None
        dw = ops.conv2d_weight_grad(x, dy, w.shape[-1], ctx.stride, pad=ctx.pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None, None, None


class _BNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, res, relu: bool, eps: float, stats: list, box=None):
        ctx.box = box  # not None: the residual's gradient goes into the box (for the unit's first conv), not back through autograd
        ctx.sync = _sync_world()
        if ctx.sync is not None:  # SyncBatchNorm: statistics over every rank's pixels
            y, mean, invstd, ctx.count = ops.sync_bn_train_forward(x, gamma, beta, eps, res, relu, *ctx.sync)
        else:
            y, mean, invstd = ops.bn_train_forward(x, gamma, beta, eps, res, relu)
            ctx.count = x.shape[0] * x.shape[2] * x.shape[3]
        # without a residual input the backward needs nothing of y: the ReLU mask is recomputed from x (hh_bn_train_backward_plain);
        # beta is saved instead (HH_TRAIN_BN_KEEP_Y=1: the stored-output form, for A/B)
        ctx.plain = res is None and ctx.sync is None and not _KEEP_Y
        ctx.save_for_backward(x, beta.detach() if ctx.plain else y, mean, invstd, gamma)
        ctx.relu, ctx.has_res = relu, res is not None
        stats.append((mean, invstd, ctx.count))
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        if ctx.sync is not None:
            dx, dgamma, dbeta, dres = ops.sync_bn_train_backward(x, y, dy.contiguous(memory_format=torch.channels_last), mean, invstd,
                                                                 gamma, ctx.relu, ctx.has_res, ctx.sync[0], ctx.count)
            if ctx.box is not None:
                _box_put(ctx.box, dres)
                dres = None
            return dx, dgamma, dbeta, dres, None, None, None, None
        if ctx.plain:  # (y holds beta)
            dx, dgamma, dbeta, dres = ops.bn_train_backward(x, None, dy.contiguous(memory_format=torch.channels_last), mean, invstd, gamma,
                                                            ctx.relu, beta=y)
        else:
            dx, dgamma, dbeta, dres = ops.bn_train_backward(x, y, dy.contiguous(memory_format=torch.channels_last), mean, invstd, gamma,
                                                            ctx.relu, want_dres=ctx.has_res)
        if ctx.box is not None:
            _box_put(ctx.box, dres)
            dres = None
        return dx, dgamma, dbeta, dres, None, None, None, None


class _FusionSumFn(torch.autograd.Function):
    """FusionLayer's `relu(sum_j f_ij(x_j))` (hrnet.py:214-229) with the nearest upsample of the low-resolution terms folded into
    the read (hh_fusion_sum_forward / _backward): no upsampled tensor, no chain of elementwise adds."""

    @staticmethod
    def forward(ctx, shifts: tuple, *terms: Tensor):
        out = ops.fusion_sum(list(terms), list(shifts), relu=True)
        ctx.save_for_backward(out)
        ctx.shifts = shifts
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        (out,) = ctx.saved_tensors
        return (None, *ops.fusion_sum_backward(dy, out, list(ctx.shifts), relu=True))


def _pad_c(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def conv(x: Tensor, m: nn.Conv2d, stride: int | None = None, box: _ResBox | None = None) -> Tensor:
    """nn.Conv2d forward on the HIP kernels.  Channel counts the kernels cannot take (3, 17, 34, 66 ...) are zero padded:
    padding and slicing are differentiable torch ops, so the gradients reach the unpadded parameter."""
    w = m.weight
    cout, cin, ks, _ = w.shape
    stride = m.stride[0] if stride is None else stride
    cin_p, cout_p = _pad_c(cin, 16), _pad_c(cout, 16)  # the data gradient runs the conv with the roles swapped
    if cin_p != cin or cout_p != cout:
        w = F.pad(w, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
    if x.shape[1] != cin_p:
        x = F.pad(x, (0, 0, 0, 0, 0, cin_p - x.shape[1]))
    pk = _PACKED[0].get((id(m.weight), stride)) if (_PACKED[0] is not None and w is m.weight) else None
    if box is not None and (cin_p != cin or x.shape[1] != cin):
        raise ValueError("conv: a skip-gradient box needs an unpadded input")
    y = _ConvFn.apply(x.contiguous(memory_format=torch.channels_last), w, stride, None, *(pk if pk is not None else (None, None)), box)
    if cout_p != cout:
        y = y[:, :cout]
    if m.bias is not None:
        y = y + m.bias.view(1, -1, 1, 1).to(y.dtype)
    return y


_PACKED: list = [None]  # {(id(weight), stride): (forward-packed, data-gradient-packed)} of the forward in flight (_refresh_packed)


def _refresh_packed(net) -> None:
    """Pack the weights of every conv whose channel counts the kernels take unpadded - forward layout and data-gradient
    layout - in ONE launch per step (ops.PackedConvWeights) instead of one launch per conv call.  The copies are read by this
    forward and the backward that follows it; they go stale with the optimizer step and are refreshed by the next forward."""
    cache = getattr(net, "_train_packed", None)
    if cache is None or not cache[0].pointers_current():
        entries, index = [], {}
        for m in net.modules():
            if isinstance(m, nn.Conv2d) and m.weight.is_cuda and m.weight.dtype == torch.float32:
                cout, cin, ks, _ = m.weight.shape
                stride = m.stride[0]
                if cin % 16 or cout % 16 or (id(m.weight), stride) in index:
                    continue
                index[(id(m.weight), stride)] = (len(entries), len(entries) + 1)
                entries += [(m.weight, stride, False), (m.weight, stride, True)]
        pw = ops.PackedConvWeights(entries)
        cache = (pw, {k: (pw.buffers[a], pw.buffers[b]) for k, (a, b) in index.items()})
        net._train_packed = cache
    cache[0].refresh()
    _PACKED[0] = cache[1]


_SYNC: list = [None]  # process group of the forward in flight when the net was converted to SyncBatchNorm (else None)


def _sync_world():
    """(group, world_size) when BatchNorm statistics are shared across ranks: like torch's SyncBatchNorm, only if a process
    group with more than one rank exists; otherwise None (plain BatchNorm)."""
    import torch.distributed as dist
    if _SYNC[0] is None or not (dist.is_available() and dist.is_initialized()):
        return None
    group = None if _SYNC[0] is True else _SYNC[0]
    world = dist.get_world_size(group)
    return (group, world) if world > 1 else None


_PENDING_STATS: list = []  # (module, batch mean, batch invstd, pixels) of the forward in flight, applied by flush_running_stats


def bn(x: Tensor, m: nn.BatchNorm2d, relu: bool = False, res: Tensor | None = None, box: _ResBox | None = None) -> Tensor:
    """nn.BatchNorm2d in training mode (+ residual, + ReLU); the running statistics are updated like torch updates them,
    once per forward for all layers together (flush_running_stats)."""
    stats: list = []
    y = _BNFn.apply(x.contiguous(memory_format=torch.channels_last), m.weight, m.bias,
                    res.contiguous(memory_format=torch.channels_last) if res is not None else None, relu, m.eps, stats, box)
    if m.track_running_stats and m.running_mean is not None:
        _PENDING_STATS.append((m, stats[0][0], stats[0][1], stats[0][2]))  # count = pixels of all ranks under SyncBatchNorm
    return y


@torch.no_grad()
def flush_running_stats() -> None:
    """running = (1 - momentum) * running + momentum * batch statistic (unbiased variance), num_batches_tracked += 1, for
    every BatchNorm of the forward in a handful of multi-tensor launches instead of six tiny ones per layer."""
    if not _PENDING_STATS:
        return
    mods = [t[0] for t in _PENDING_STATS]
    moms = {(m.momentum if m.momentum is not None else 0.1) for m in mods}
    means = [t[1] for t in _PENDING_STATS]
    # unbiased batch variance = (1 / invstd^2 - eps) * n / (n - 1), for all layers in four multi-tensor launches
    invstds = [t[2] for t in _PENDING_STATS]
    var_unb = torch._foreach_reciprocal(torch._foreach_mul(invstds, invstds))
    torch._foreach_sub_(var_unb, [float(t[0].eps) for t in _PENDING_STATS])
    torch._foreach_mul_(var_unb, [float(t[3]) / max(float(t[3]) - 1.0, 1.0) for t in _PENDING_STATS])
    if len(moms) == 1:
        mom = moms.pop()
        rm, rv = [m.running_mean for m in mods], [m.running_var for m in mods]
        torch._foreach_mul_(rm, 1 - mom); torch._foreach_add_(rm, means, alpha=mom)
        torch._foreach_mul_(rv, 1 - mom); torch._foreach_add_(rv, var_unb, alpha=mom)
    else:
        for m, mean, vu in zip(mods, means, var_unb):
            mom = m.momentum if m.momentum is not None else 0.1
            m.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            m.running_var.mul_(1 - mom).add_(vu, alpha=mom)
    torch._foreach_add_([m.num_batches_tracked for m in mods], 1)
    _PENDING_STATS.clear()


def deconv_k4s2(x: Tensor, m: nn.ConvTranspose2d) -> Tensor:
    """ConvTranspose2d(k=4, s=2, p=1, bias=False) (higher_hrnet.py:21-24) as four output-parity 2x2 convolutions: output
    row 2i+py takes input rows (i-1, i) with kernel rows (3, 1) when py = 0 and rows (i, i+1) with kernel rows (2, 0) when
    py = 1 (same in x).  The phase kernels are index views of the parameter, so autograd assembles its gradient."""
    wt = m.weight  # [cin, cout, 4, 4]
    cin, cout = wt.shape[:2]
    cin_p, cout_p = _pad_c(cin, 16), _pad_c(cout, 16)  # the data gradient runs the conv with the roles swapped
    if x.shape[1] != cin_p:
        x = F.pad(x, (0, 0, 0, 0, 0, cin_p - x.shape[1]))
    x = x.contiguous(memory_format=torch.channels_last)
    B, _, H, W = x.shape
    y = torch.empty((B, cout, 2 * H, 2 * W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)  # every phase is written
    for py in range(2):
        for px in range(2):
            ky = (3, 1) if py == 0 else (2, 0)
            kx = (3, 1) if px == 0 else (2, 0)
            # (slices + stack, not list indexing: no index tensor is uploaded, so the step can be captured in a hipGraph)
            w = torch.stack([torch.stack([wt[:, :, a, b] for b in kx], -1) for a in ky], -2).permute(1, 0, 2, 3)  # [cout, cin, 2, 2]
            w = F.pad(w, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
            yp = _ConvFn.apply(x, w.contiguous(), 1, (1 - py, 1 - px))[:, :cout]
            y[:, :, py::2, px::2] = yp
    return y


# ------------------------------------------------------------------------------------------ the net
_NO_RESBOX = bool(os.environ.get("HH_TRAIN_NO_RESBOX"))  # A/B: the skip gradients through autograd's own accumulation
_KEEP_Y = bool(os.environ.get("HH_TRAIN_BN_KEEP_Y"))  # A/B: every BatchNorm backward reads its stored output


def _boxable(x, c: nn.Conv2d) -> bool:
    return not _NO_RESBOX and x.requires_grad and c.weight.shape[1] % 16 == 0 and x.shape[1] == c.weight.shape[1]


def _bottleneck(x, u):
    ds = u._modules.get("downsample")
    box = _ResBox() if ds is None and _boxable(x, u.conv1) else None  # identity skip: its gradient joins conv1's data gradient
    y = bn(conv(x, u.conv1, box=box), u.bn1, relu=True)
    y = bn(conv(y, u.conv2), u.bn2, relu=True)
    r = bn(conv(x, ds._modules["0"]), ds._modules["1"]) if ds is not None else x
    return bn(conv(y, u.conv3), u.bn3, relu=True, res=r, box=box)


def _basic(x, u):
    box = _ResBox() if _boxable(x, u.conv1) else None
    y = bn(conv(x, u.conv1, box=box), u.bn1, relu=True)
    return bn(conv(y, u.conv2), u.bn2, relu=True, res=x, box=box)


def _children(m):
    return [m._modules[k] for k in sorted(m._modules, key=int)]


def _fusion(xs, fl, n_out):
    outs = []
    for i in range(n_out):
        row = fl.scales_fusion_layers._modules.get(str(i)) if hasattr(fl, "scales_fusion_layers") else None
        terms, shifts = [xs[i]], [0]  # the identity term first: it has the output resolution
        for j, x in enumerate(xs):
            if j == i:
                continue
            q = row._modules[str(j)]
            if j > i:  # 1x1 conv + BN at the low resolution; nn.Upsample(nearest) happens inside the sum
                terms.append(bn(conv(x, q._modules["0"]), q._modules["1"]))
                shifts.append(j - i)
            else:
                t = x
                for k in range(i - j):
                    qq = q._modules[str(k)]
                    t = bn(conv(t, qq._modules["0"]), qq._modules["1"], relu=(k != i - j - 1))
                terms.append(t)
                shifts.append(0)
        outs.append(_FusionSumFn.apply(tuple(shifts), *terms))
    return outs


def higher_hrnet_train_forward(net, images: Tensor):
    """-> ([hm_1/4, hm_1/2] fp32, tags_1/4 fp32), differentiable w.r.t. every parameter of `net`."""
    K = net.num_kpts
    bb = net.backbone
    _PENDING_STATS.clear()
    _SYNC[0] = getattr(net, "sync_batchnorm", None)  # set by KeypointsModel.to_DDP(..., use_batchnorm=True)
    _refresh_packed(net)
    x = images.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = bn(conv(x, bb.conv1), bb.bn1, relu=True)
    x = bn(conv(x, bb.conv2), bb.bn2, relu=True)
    xs = [x]
    nblocks = [1, 1, 4, 3]
    for s in range(4):
        st = bb.stages._modules[str(s)]
        for b in range(nblocks[s]):
            blk = st.blocks._modules[str(2 * b)]
            unit = _bottleneck if s == 0 else _basic
            new = []
            for i, t in enumerate(xs):
                for u in _children(blk.scales_blocks._modules[str(i)]):
                    t = unit(t, u)
                new.append(t)
            xs = new
            last = s == 3 and b == nblocks[s] - 1
            if s > 0:
                xs = _fusion(xs, st.blocks._modules[str(2 * b + 1)], 1 if last else len(xs))
            # (stage 0 has one scale: its "fusion" is the ReLU of a ReLU output, hrnet.py:221-229 -- the identity, gradient included)
        if s < 3:
            tb = st.transition_layer.transition_blocks
            n = len(xs)
            q = tb._modules[str(n)]
            newb = bn(conv(xs[-1], q._modules["0"]), q._modules["1"], relu=True)
            if s == 0:
                q0 = tb._modules["0"]
                xs = [bn(conv(xs[0], q0._modules["0"]), q0._modules["1"], relu=True)]
            xs = xs + [newb]
    feats = xs[0]
    init = conv(feats, net.init_heatmaps_head)
    d = net.deconv_layers._modules["0"]
    y = torch.cat((feats, init.to(feats.dtype)), 1)
    y = bn(deconv_k4s2(y, d.deconv._modules["0"]), d.deconv._modules["1"], relu=True)
    for u in _children(d.resid_blocks):
        y = _basic(y, u)
    out = conv(y, d.final_layer)
    flush_running_stats()
    init, out = init.float(), out.float()
    return [init[:, :K], out[:, :K]], init[:, K:]
