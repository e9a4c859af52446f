"""Parameter layout of HRNet / HigherHRNet expressed as a flat table.

The drop-in contract is the reference's *state-dict key names*
(`/root/reference/src/keypoints/architectures/hrnet.py:29-385`,
`higher_hrnet.py:7-64`): 1810 keys for W32.  The compute graph itself lives in the HIP
engine (`csrc/engine.cpp`); Python only needs containers that own parameters under the same
names, so this module enumerates `(dotted_path, kind, ctor_args)` rows and
`attach_modules` hangs real `nn.Conv2d` / `nn.BatchNorm2d` / `nn.ConvTranspose2d`
leaves at those paths (so `.modules()`-based initialisers such as
`src/keypoints/model.py:19-34` keep working).
"""
from __future__ import annotations

from typing import Iterator

from torch import nn

Row = tuple[str, str, tuple]


def _conv(path, cin, cout, k, s, bias=False) -> Row:
    return (path, "conv", (cin, cout, k, s, (k - 1) // 2, bias))


def _bn(path, c) -> Row:
    return (path, "bn", (c,))


def backbone_rows(C: int, prefix: str = "backbone", single_scale_out: bool = True) -> Iterator[Row]:
    """Rows for HRNetBackbone (hrnet.py:342-385)."""
    p = prefix
    yield _conv(f"{p}.conv1", 3, 64, 3, 2)
    yield _bn(f"{p}.bn1", 64)
    yield _conv(f"{p}.conv2", 64, 64, 3, 2)
    yield _bn(f"{p}.bn2", 64)
    widths = [C, 2 * C, 4 * C, 8 * C]
    # (num_blocks, unit kind, branch widths entering the stage) -- hrnet.py:347-353
    stages = [(1, "bottleneck", [64]), (1, "basic", widths[:2]), (4, "basic", widths[:3]), (3, "basic", widths[:4])]
    for s, (nblocks, unit, cin_list) in enumerate(stages):
        sp = f"{p}.stages.{s}"
        cin_list = list(cin_list)
        for b in range(nblocks):
            hp = f"{sp}.blocks.{2 * b}"  # HR block at even index, fusion at odd (hrnet.py:319-323)
            for i, cin in enumerate(cin_list):
                for u in range(4):
                    up = f"{hp}.scales_blocks.{i}.{u}"
                    if unit == "bottleneck":
                        cu_in = cin if u == 0 else cin * 4
                        cout, mid = cin * 4, cin
                        yield _conv(f"{up}.conv1", cu_in, mid, 1, 1)
                        yield _bn(f"{up}.bn1", mid)
                        yield _conv(f"{up}.conv2", mid, mid, 3, 1)
                        yield _bn(f"{up}.bn2", mid)
                        yield _conv(f"{up}.conv3", mid, cout, 1, 1)
                        yield _bn(f"{up}.bn3", cout)
                        if cu_in != cout:
                            yield _conv(f"{up}.downsample.0", cu_in, cout, 1, 1)
                            yield _bn(f"{up}.downsample.1", cout)
                    else:
                        yield _conv(f"{up}.conv1", cin, cin, 3, 1)
                        yield _bn(f"{up}.bn1", cin)
                        yield _conv(f"{up}.conv2", cin, cin, 3, 1)
                        yield _bn(f"{up}.bn2", cin)
            # fusion (hrnet.py:166-229). Stage 0 fuses a single scale: identity, no params.
            fw = widths[: len(cin_list)]
            last = s == 3 and b == nblocks - 1
            n_out = 1 if (last and single_scale_out) else len(fw)
            fp = f"{sp}.blocks.{2 * b + 1}"
            if s > 0:
                for i in range(n_out):
                    for j in range(len(fw)):
                        lp = f"{fp}.scales_fusion_layers.{i}.{j}"
                        if j > i:  # low -> high: 1x1 conv + BN + nearest upsample
                            yield _conv(f"{lp}.0", fw[j], fw[i], 1, 1)
                            yield _bn(f"{lp}.1", fw[i])
                        elif j < i:  # high -> low: chain of stride-2 3x3 convs
                            for k in range(i - j):
                                co = fw[i] if k == i - j - 1 else fw[j]
                                yield _conv(f"{lp}.{k}.0", fw[j], co, 3, 2)
                                yield _bn(f"{lp}.{k}.1", co)
        if s < 3:  # transition (hrnet.py:232-284)
            tp = f"{sp}.transition_layer.transition_blocks"
            if s == 0:
                yield _conv(f"{tp}.0.0", 256, widths[0], 3, 1)
                yield _bn(f"{tp}.0.1", widths[0])
                yield _conv(f"{tp}.1.0", 256, widths[1], 3, 2)
                yield _bn(f"{tp}.1.1", widths[1])
            else:
                n = len(cin_list)
                yield _conv(f"{tp}.{n}.0", widths[n - 1], widths[n], 3, 2)
                yield _bn(f"{tp}.{n}.1", widths[n])


def higher_hrnet_rows(num_kpts: int, C: int) -> Iterator[Row]:
    """Rows for HigherHRNet (higher_hrnet.py:47-64)."""
    yield from backbone_rows(C, "backbone", True)
    K = num_kpts
    yield _conv("init_heatmaps_head", C, 2 * K, 1, 1, True)
    dp = "deconv_layers.0"
    yield (f"{dp}.deconv.0", "deconv", (C + 2 * K, C, 4, 2, 1, 0))
    yield _bn(f"{dp}.deconv.1", C)
    for r in range(4):
        rp = f"{dp}.resid_blocks.{r}"
        yield _conv(f"{rp}.conv1", C, C, 3, 1)
        yield _bn(f"{rp}.bn1", C)
        yield _conv(f"{rp}.conv2", C, C, 3, 1)
        yield _bn(f"{rp}.bn2", C)
    yield _conv(f"{dp}.final_layer", C, K, 1, 1, True)


def classification_hrnet_rows(C: int, num_classes: int = 1000) -> Iterator[Row]:
    """Rows for ClassificationHRNet (classification/architectures/hrnet.py:7-74): 4-scale backbone + head."""
    yield from backbone_rows(C, "backbone", False)
    widths, outs = [C, 2 * C, 4 * C, 8 * C], [128, 256, 512, 1024]
    hp = "classification_head"
    for i in range(4):
        up, cin, cout, mid = f"{hp}.chann_incr_blocks.{i}", widths[i], outs[i], outs[i] // 4
        yield _conv(f"{up}.conv1", cin, mid, 1, 1)
        yield _bn(f"{up}.bn1", mid)
        yield _conv(f"{up}.conv2", mid, mid, 3, 1)
        yield _bn(f"{up}.bn2", mid)
        yield _conv(f"{up}.conv3", mid, cout, 1, 1)
        yield _bn(f"{up}.bn3", cout)
        if cin != cout:
            yield _conv(f"{up}.downsample.0", cin, cout, 1, 1)
            yield _bn(f"{up}.downsample.1", cout)
    for i in range(3):
        yield _conv(f"{hp}.downsample_blocks.{i}.0", outs[i], outs[i + 1], 3, 2, True)
        yield _bn(f"{hp}.downsample_blocks.{i}.1", outs[i + 1])
    yield _conv(f"{hp}.final_conv.0", 1024, 2048, 1, 1, True)
    yield _bn(f"{hp}.final_conv.1", 2048)
    yield (f"{hp}.classifier", "linear", (2048, num_classes))


class _Node(nn.Module):
    """Anonymous container; children are addressed by name only."""


def attach_modules(root: nn.Module, rows) -> None:
    for path, kind, a in rows:
        parts = path.split(".")
        node = root
        for name in parts[:-1]:
            child = node._modules.get(name)
            if child is None:
                child = _Node()
                node.add_module(name, child)
            node = child
        if kind == "conv":
            cin, cout, k, s, pad, bias = a
            leaf = nn.Conv2d(cin, cout, k, s, pad, bias=bias)
        elif kind == "bn":
            leaf = nn.BatchNorm2d(a[0])
        elif kind == "deconv":
            cin, cout, k, s, pad, opad = a
            leaf = nn.ConvTranspose2d(cin, cout, k, s, pad, opad, bias=False)
        elif kind == "linear":
            leaf = nn.Linear(a[0], a[1])
        else:  # pragma: no cover
            raise ValueError(kind)
        node.add_module(parts[-1], leaf)
