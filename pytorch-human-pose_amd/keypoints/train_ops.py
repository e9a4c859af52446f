"""Python face of the training building blocks (`hh_conv2d`, `hh_bn_train_*`): NHWC bf16 activations as torch tensors
in channels_last memory format, fp32 parameters.  `train_net.py` assembles them into the net's training forward."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _lib


def _nhwc(x: Tensor) -> Tensor:
    """[B,C,H,W] bf16 tensor whose memory is NHWC (channels_last); returns it contiguous in that format."""
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4:
        raise _lib.HHError("expected a 4-d CUDA/HIP bfloat16 tensor: there is no CPU path")
    return x.contiguous(memory_format=torch.channels_last)


def conv2d(x: Tensor, w: Tensor, stride: int = 1, bias: Tensor | None = None, res: Tensor | None = None, relu: bool = False,
           data_grad: bool = False, pad: tuple[int, int] | None = None, packed: Tensor | None = None) -> Tensor:
    """y = act(conv(x, w) + bias (+ res)), padding (ks-1)/2.  data_grad=True: x is dL/dy of the conv with weights
    w [cout,cin,ks,ks] and this stride, and the result is dL/dx (stride 2: 3x3 only, even input sizes).
    packed: the weights of this (w, stride, data_grad) already packed by `pack_conv_weights` (w then only gives the shape)."""
    lib = _lib.load()
    x = _nhwc(x)
    B, Cx, H, W = x.shape
    cout, cin, ks, _ = w.shape
    if packed is not None:
        if Cx != (cout if data_grad else cin):
            raise ValueError(f"conv2d: input has {Cx} channels, weights {tuple(w.shape)}, data_grad={data_grad}")
        mode = (2 if stride == 2 else 1) if data_grad else 0
        co = cin if data_grad else cout
        Ho, Wo = (2 * H, 2 * W) if mode == 2 else ((H // 2, W // 2) if stride == 2 else (H, W))
        y = torch.empty((B, co, Ho, Wo), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        if res is not None:
            res = _nhwc(res)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        py_, px_ = pad if pad is not None else (-1, -1)
        with torch.cuda.device(x.device):
            _lib.check(lib.hh_conv2d_packed(x.data_ptr(), B, H, W, cin, packed.data_ptr(), cout, ks, stride, mode, py_, px_,
                                            bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None,
                                            int(relu), y.data_ptr(), stream))
        return y
    w = w.detach().to(x.device, torch.float32).contiguous()
    if Cx != (cout if data_grad else cin):
        raise ValueError(f"conv2d: input has {Cx} channels, weights {tuple(w.shape)}, data_grad={data_grad}")
    co = cin if data_grad else cout
    mode = (2 if stride == 2 else 1) if data_grad else 0
    if mode == 2:
        Ho, Wo = 2 * H, 2 * W  # dL/dx of a stride-2 conv lives on the input grid
    else:
        Ho, Wo = (H // 2, W // 2) if stride == 2 else (H, W)
    y = torch.empty((B, co, Ho, Wo), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    nbytes = lib.hh_conv2d_workspace_bytes(cin, cout, ks, mode)
    if nbytes < 0:
        raise _lib.HHError("conv2d: no kernel family for this shape")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    if res is not None:
        res = _nhwc(res)
    if bias is not None:
        bias = bias.detach().to(x.device, torch.float32).contiguous()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        py_, px_ = pad if pad is not None else (-1, -1)
        _lib.check(lib.hh_conv2d(x.data_ptr(), B, H, W, cin, w.data_ptr(), cout, ks, stride, mode, py_, px_,
                                 bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None,
                                 int(relu), y.data_ptr(), ws.data_ptr(), stream))
    return y


def bn_train_forward(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, res: Tensor | None = None, relu: bool = False):
    """-> (y, mean, invstd): BatchNorm2d in training mode (+ residual, + ReLU); mean / invstd feed the backward."""
    lib = _lib.load()
    x = _nhwc(x)
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C, device=x.device, dtype=torch.float32)
    scratch = torch.empty(256 * C * 2, device=x.device, dtype=torch.float64)
    if res is not None:
        res = _nhwc(res)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(lib.hh_bn_train_forward(x.data_ptr(), B * H * W, C, gamma.float().contiguous().data_ptr(), beta.float().contiguous().data_ptr(),
                                           eps, res.data_ptr() if res is not None else None, int(relu), y.data_ptr(), mean.data_ptr(),
                                           invstd.data_ptr(), scratch.data_ptr(), stream))
    return y, mean, invstd


def bn_train_backward(x: Tensor, y, dy: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, relu: bool = False,
                      want_dres: bool = False, beta=None):
    """-> (dx, dgamma, dbeta, dres or None).  y = None (with beta): a BatchNorm without a residual input, the ReLU mask is
    recomputed from x (hh_bn_train_backward_plain)."""
    lib = _lib.load()
    if y is None:
        assert beta is not None and not want_dres
        x, dy = _nhwc(x), _nhwc(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
        scratch = torch.empty(256 * C * 2, device=x.device, dtype=torch.float64)
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(lib.hh_bn_train_backward_plain(x.data_ptr(), dy.data_ptr(), B * H * W, C, mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                                      b.data_ptr(), int(relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                      scratch.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
        return dx, dgamma, dbeta, None
    x, y, dy = _nhwc(x), _nhwc(y), _nhwc(dy)
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
    scratch = torch.empty(256 * C * 2, device=x.device, dtype=torch.float64)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    g = gamma.float().contiguous()
    with torch.cuda.device(x.device):
        _lib.check(lib.hh_bn_train_backward(x.data_ptr(), y.data_ptr(), dy.data_ptr(), B * H * W, C, mean.data_ptr(), invstd.data_ptr(),
                                            g.data_ptr(), int(relu), dx.data_ptr(), dres.data_ptr() if dres is not None else None,
                                            dgamma.data_ptr(), dbeta.data_ptr(), scratch.data_ptr(), stream))
    return dx, dgamma, dbeta, dres


class PackedConvWeights:
    """bf16 kernel-layout copies of a set of conv weights, all refreshed by ONE launch (`refresh`, once per training step
    before the forward).  entries: (weight [cout,cin,ks,ks] fp32 CUDA parameter, stride, data_grad)."""

    def __init__(self, entries: list):
        import ctypes as C
        lib = _lib.load()
        self.n = len(entries)
        self.buffers: list[Tensor] = []
        self._keep = [w for w, _, _ in entries]
        shapes = (C.c_int32 * (5 * self.n))()
        dev = entries[0][0].device if entries else None
        for i, (w, stride, data_grad) in enumerate(entries):
            cout, cin, ks, _ = w.shape
            mode = (2 if stride == 2 else 1) if data_grad else 0
            nel = lib.hh_conv2d_packed_elems(cin, cout, ks, stride, mode)
            if nel < 0:
                raise _lib.HHError(f"no kernel family for conv weights {tuple(w.shape)} (stride {stride}, data_grad={data_grad})")
            self.buffers.append(torch.empty(nel, device=dev, dtype=torch.bfloat16))
            shapes[5 * i:5 * i + 5] = [cout, cin, ks, stride, mode]
        self._shapes = shapes
        self._w = (C.c_void_p * self.n)(*[w.data_ptr() for w, _, _ in entries])
        self._p = (C.c_void_p * self.n)(*[b.data_ptr() for b in self.buffers])
        self._descs = torch.empty(max(1, self.n) * 4 * 64, device=dev, dtype=torch.uint8) if entries else None

    def pointers_current(self) -> bool:
        return all(w.data_ptr() == p for w, p in zip(self._keep, self._w))

    def refresh(self) -> None:
        if not self.n:
            return
        lib = _lib.load()
        dev = self._descs.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(lib.hh_pack_conv_weights_batch(self.n, self._w, self._p, self._shapes, self._descs.data_ptr(), stream))


def _all_reduce_sums(sums: Tensor, group) -> None:
    import torch.distributed as dist
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)  # RCCL on the GPU box; 2C doubles per layer


def sync_bn_train_forward(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, res: Tensor | None, relu: bool, group, world: int):
    """SyncBatchNorm forward (base/model.py:42-44): statistics over the pixels of ALL ranks of `group`.  Every rank must hold
    the same number of pixels (DistributedSampler with drop_last=True, datamodule.py:68-89).  -> (y, mean, invstd, count)"""
    lib = _lib.load()
    x = _nhwc(x)
    B, C, H, W = x.shape
    P = B * H * W
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    invstd = torch.empty(C, device=x.device, dtype=torch.float32)
    sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
    scratch = torch.empty(256 * C * 2, device=x.device, dtype=torch.float64)
    if res is not None:
        res = _nhwc(res)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(lib.hh_bn_train_stats(x.data_ptr(), P, C, sums.data_ptr(), scratch.data_ptr(), stream))
        _all_reduce_sums(sums, group)
        count = float(P) * world
        _lib.check(lib.hh_bn_train_normalize(x.data_ptr(), P, C, sums.data_ptr(), count, gamma.float().contiguous().data_ptr(),
                                             beta.float().contiguous().data_ptr(), eps, res.data_ptr() if res is not None else None,
                                             int(relu), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), stream))
    return y, mean, invstd, count


def sync_bn_train_backward(x: Tensor, y: Tensor, dy: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, relu: bool, want_dres: bool,
                           group, count: float):
    """-> (dx, dgamma, dbeta, dres or None); dgamma / dbeta are this rank's sums (DDP averages parameter gradients)."""
    lib = _lib.load()
    x, y, dy = _nhwc(x), _nhwc(y), _nhwc(dy)
    B, C, H, W = x.shape
    P = B * H * W
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
    sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
    scratch = torch.empty(256 * C * 2, device=x.device, dtype=torch.float64)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    g = gamma.float().contiguous()
    with torch.cuda.device(x.device):
        _lib.check(lib.hh_bn_train_backward_stats(x.data_ptr(), y.data_ptr(), dy.data_ptr(), P, C, mean.data_ptr(), invstd.data_ptr(),
                                                  int(relu), sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), scratch.data_ptr(), stream))
        _all_reduce_sums(sums, group)
        _lib.check(lib.hh_bn_train_backward_apply(x.data_ptr(), y.data_ptr(), dy.data_ptr(), P, C, mean.data_ptr(), invstd.data_ptr(),
                                                  g.data_ptr(), int(relu), sums.data_ptr(), count, dx.data_ptr(),
                                                  dres.data_ptr() if dres is not None else None, scratch.data_ptr(), stream))
    return dx, dgamma, dbeta, dres


def conv2d_weight_grad(x: Tensor, dy: Tensor, ks: int, stride: int = 1, pad: tuple[int, int] | None = None) -> Tensor:
    """dL/dW [cout,cin,ks,ks] fp32 of y = conv(x, W) (padding (ks-1)/2) from the layer input x and dL/dy."""
    lib = _lib.load()
    x, dy = _nhwc(x), _nhwc(dy)
    B, cin, H, W = x.shape
    cout = dy.shape[1]
    dw = torch.empty((cout, cin, ks, ks), device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.hh_conv2d_wgrad_workspace_bytes(B, H, W, cin, cout, ks, stride), device=x.device, dtype=torch.uint8)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        py_, px_ = pad if pad is not None else (-1, -1)
        _lib.check(lib.hh_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), B, H, W, cin, cout, ks, stride, py_, px_, dw.data_ptr(), ws.data_ptr(), stream))
    return dw


def fusion_sum(terms: list[Tensor], shifts: list[int], relu: bool = True) -> Tensor:
    """out = act(sum_j nearest_upsample(terms[j], 2 ** shifts[j])) (hh_fusion_sum_forward); terms[0] has the output resolution."""
    import ctypes as C
    lib = _lib.load()
    terms = [_nhwc(t) for t in terms]
    B, Cc, H, W = terms[0].shape
    out = torch.empty_like(terms[0])
    ptrs = (C.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    sh = (C.c_int * len(terms))(*shifts)
    stream = torch.cuda.current_stream(out.device).cuda_stream
    with torch.cuda.device(out.device):
        _lib.check(lib.hh_fusion_sum_forward(ptrs, sh, len(terms), B, H, W, Cc, int(relu), out.data_ptr(), stream))
    return out


def fusion_sum_backward(dy: Tensor, out: Tensor, shifts: list[int], relu: bool = True) -> list[Tensor]:
    """-> the gradient of every term of fusion_sum (shift-0 terms share one tensor)."""
    import ctypes as C
    lib = _lib.load()
    dy, out = _nhwc(dy), _nhwc(out)
    B, Cc, H, W = dy.shape
    g = torch.empty_like(dy) if relu else dy
    ups = [(j, s) for j, s in enumerate(shifts) if s > 0]
    dups = [torch.empty((B, Cc, H >> s, W >> s), device=dy.device, dtype=dy.dtype).contiguous(memory_format=torch.channels_last) for _, s in ups]
    ptrs = (C.c_void_p * max(len(dups), 1))(*[t.data_ptr() for t in dups])
    sh = (C.c_int * max(len(dups), 1))(*[s for _, s in ups])
    stream = torch.cuda.current_stream(dy.device).cuda_stream
    with torch.cuda.device(dy.device):
        _lib.check(lib.hh_fusion_sum_backward(dy.data_ptr(), out.data_ptr(), int(relu), B, H, W, Cc, g.data_ptr() if relu else None, ptrs, sh,
                                              len(dups), stream))
    grads: list = [g] * len(shifts)
    for (j, _), d in zip(ups, dups):
        grads[j] = d
    return grads
