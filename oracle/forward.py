"""CPU fp32 restatement of the HigherHRNet forward (torch functional ops over a state dict).

TEST INFRASTRUCTURE ONLY (checker for the HIP engine; also bench.py's cpu_baseline leg).
Parity status: PINNED by tests/golden/net_forward.npz (tools/make_golden.py imports the
reference's HigherHRNet in the build container and stores its outputs on this repo's
seeded synthetic weights/inputs).

The walk below follows the reference module tree purely through the state-dict key names
(SURVEY.md §8b); each helper cites the reference lines it restates
(paths relative to /root/reference/src/keypoints/architectures/).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5


class _SD:
    def __init__(self, sd: dict, prefix: str = ""):
        self.sd, self.prefix = sd, prefix

    def sub(self, name: str) -> "_SD":
        return _SD(self.sd, f"{self.prefix}{name}.")

    def has(self, name: str) -> bool:
        return f"{self.prefix}{name}" in self.sd

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.sd[f"{self.prefix}{name}"]


_TRAIN = False  # set by higher_hrnet(..., train=True): BatchNorm normalises with batch statistics (nn.Module.train())


def _bn(x, p: _SD):
    if _TRAIN:  # hrnet.py uses nn.BatchNorm2d defaults: batch statistics, momentum 0.1 (running stats not needed here)
        return F.batch_norm(x, None, None, p["weight"], p["bias"], True, 0.0, EPS)
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], False, 0.0, EPS)


def _conv_bn(x, p: _SD, conv: str, bn: str, stride=1, pad=None, relu=False):
    w = p[f"{conv}.weight"]
    if pad is None:
        pad = (w.shape[-1] - 1) // 2
    y = _bn(F.conv2d(x, w, None, stride, pad), p.sub(bn))
    return F.relu(y) if relu else y


def bottleneck(x, p: _SD):
    """hrnet.py:58-74"""
    y = _conv_bn(x, p, "conv1", "bn1", relu=True)
    y = _conv_bn(y, p, "conv2", "bn2", relu=True)
    y = _conv_bn(y, p, "conv3", "bn3")
    r = _conv_bn(x, p, "downsample.0", "downsample.1") if p.has("downsample.0.weight") else x
    return F.relu(y + r)


def basic_block(x, p: _SD):
    """hrnet.py:108-124"""
    y = _conv_bn(x, p, "conv1", "bn1", relu=True)
    y = _conv_bn(y, p, "conv2", "bn2")
    return F.relu(y + x)


def hr_block(xs, p: _SD, unit):
    """hrnet.py:154-163: four residual units per scale, scales independent."""
    out = []
    for i, x in enumerate(xs):
        for u in range(4):
            x = unit(x, p.sub(f"scales_blocks.{i}.{u}"))
        out.append(x)
    return out


def fusion(xs, p: _SD, n_out: int):
    """hrnet.py:214-229"""
    outs = []
    for i in range(n_out):
        acc = 0
        for j, x in enumerate(xs):
            q = p.sub(f"scales_fusion_layers.{i}.{j}")
            if j == i:
                t = x
            elif j > i:  # 1x1 conv + BN + nearest upsample (hrnet.py:200-205)
                t = _conv_bn(x, q, "0", "1")
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:  # chain of stride-2 3x3 convs, ReLU on all but the last (hrnet.py:183-197)
                t = x
                for k in range(i - j):
                    t = _conv_bn(t, q.sub(str(k)), "0", "1", stride=2, relu=(k != i - j - 1))
            acc = acc + t
        outs.append(F.relu(acc))
    return outs


def backbone(x, p: _SD, single_scale_out: bool = True):
    """hrnet.py:378-385 + stages (hrnet.py:333-339) + transitions (hrnet.py:270-284)."""
    x = _conv_bn(x, p, "conv1", "bn1", stride=2, relu=True)
    x = _conv_bn(x, p, "conv2", "bn2", stride=2, relu=True)
    taps = {"stem#0": x}
    xs = [x]
    nblocks = [1, 1, 4, 3]
    for s in range(4):
        sp = p.sub(f"stages.{s}")
        for b in range(nblocks[s]):
            xs = hr_block(xs, sp.sub(f"blocks.{2 * b}"), bottleneck if s == 0 else basic_block)
            for i, t in enumerate(xs):
                taps[f"stages.{s}.blocks.{2 * b}#{i}"] = t
            last = s == 3 and b == nblocks[s] - 1
            n_out = 1 if (last and single_scale_out) else len(xs)
            xs = fusion(xs, sp.sub(f"blocks.{2 * b + 1}"), n_out) if s > 0 else [F.relu(xs[0])]
            for i, t in enumerate(xs):
                taps[f"stages.{s}.blocks.{2 * b + 1}#{i}"] = t
        if s < 3:
            tp = sp.sub("transition_layer.transition_blocks")
            n = len(xs)
            new = _conv_bn(xs[-1], tp.sub(str(n)), "0", "1", stride=2, relu=True)
            if s == 0:
                xs = [_conv_bn(xs[0], tp.sub("0"), "0", "1", relu=True)]
            xs = xs + [new]
        for i, t in enumerate(xs):
            taps[f"stages.{s}#{i}"] = t
    return xs, taps


def higher_hrnet(images: torch.Tensor, sd: dict, num_kpts: int = 17, return_taps: bool = False, train: bool = False):
    """higher_hrnet.py:66-81 -> ([hm_1/4, hm_1/2], tags_1/4).  train=True: the net in .train() mode (batch-stat BN)."""
    global _TRAIN
    _TRAIN = train
    try:
        return _higher_hrnet(images, sd, num_kpts, return_taps)
    finally:
        _TRAIN = False


def _higher_hrnet(images: torch.Tensor, sd: dict, num_kpts: int, return_taps: bool):
    p = _SD(sd)
    K = num_kpts
    xs, taps = backbone(images, p.sub("backbone"), True)
    feats = xs[0]
    init = F.conv2d(feats, p["init_heatmaps_head.weight"], p["init_heatmaps_head.bias"])
    d = p.sub("deconv_layers.0")
    y = torch.cat((feats, init), 1)
    y = F.conv_transpose2d(y, d["deconv.0.weight"], None, 2, 1, 0)
    y = F.relu(_bn(y, d.sub("deconv.1")))
    for r in range(4):
        y = basic_block(y, d.sub(f"resid_blocks.{r}"))
    taps["deconv#0"] = y
    out = F.conv2d(y, d["final_layer.weight"], d["final_layer.bias"])
    taps["deconv#1"] = out
    hms, tags = [init[:, :K], out[:, :K]], init[:, K:]
    return (hms, tags, taps) if return_taps else (hms, tags)


def classification_hrnet(images: torch.Tensor, sd: dict) -> torch.Tensor:
    """ClassificationHRNet.forward (/root/reference/src/classification/architectures/hrnet.py:48-74): 4-scale backbone,
    one Bottleneck per scale (C_i -> 128/256/512/1024), stride-2 conv(+bias)+BN+ReLU downsample-and-add chain,
    1x1 -> 2048 + BN + ReLU, global average pool, Linear."""
    p = _SD(sd)
    xs, _ = backbone(images, p.sub("backbone"), single_scale_out=False)
    hp = p.sub("classification_head")
    out = bottleneck(xs[0], hp.sub("chann_incr_blocks.0"))
    for i in range(3):
        d = hp.sub(f"downsample_blocks.{i}")
        down = F.relu(_bn(F.conv2d(out, d["0.weight"], d["0.bias"], 2, 1), d.sub("1")))
        out = bottleneck(xs[i + 1], hp.sub(f"chann_incr_blocks.{i + 1}")) + down
    f = hp.sub("final_conv")
    out = F.relu(_bn(F.conv2d(out, f["0.weight"], f["0.bias"]), f.sub("1")))
    flat = F.avg_pool2d(out, kernel_size=out.shape[2:]).view(out.shape[0], -1)
    return F.linear(flat, hp["classifier.weight"], hp["classifier.bias"])


COCO_FLIP_INDEX = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]  # keypoints/transforms.py:11


def flip_tta(images: torch.Tensor, sd: dict, num_kpts: int = 17):
    """keypoints/model.py:85-94 -> ([hm_1/4, hm_1/2] averaged, [tags, tags_flipped])."""
    hms, tags = higher_hrnet(images, sd, num_kpts)
    fh, ft = higher_hrnet(torch.flip(images, [3]), sd, num_kpts)
    hms = [(hms[i] + torch.flip(fh[i], [3])[:, COCO_FLIP_INDEX]) / 2 for i in range(2)]
    return hms, [tags, torch.flip(ft, [3])[:, COCO_FLIP_INDEX]]
