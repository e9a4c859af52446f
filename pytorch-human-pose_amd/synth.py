"""Deterministic synthetic weights / inputs for the HigherHRNet hot path.

There is no COCO checkpoint in the build environment (SURVEY.md §7 "Hard parts"), so every
parity test, golden fixture and the benchmark run on seeded synthetic parameters.  The
generator is keyed on the *state-dict key name* (the drop-in contract of SURVEY.md §8b), so
the reference module tree (golden generation, this container only) and this package's
module tree receive bit-identical fp32 parameters without sharing any code.

numpy ``RandomState`` (MT19937) streams are stable across numpy versions, so fixtures only
need to store outputs, never weights.
"""
from __future__ import annotations

import re
import zlib

import numpy as np

_LAST_BN = re.compile(
    r"(stages\.0\.blocks\.\d+\.scales_blocks\.\d+\.\d+\.bn3|"  # Bottleneck tail (hrnet.py:47-48)
    r"stages\.[123]\.blocks\.\d+\.scales_blocks\.\d+\.\d+\.bn2|"  # BasicBlock tail (hrnet.py:94-100)
    r"resid_blocks\.\d+\.bn2)\."
)
_FUSION_BN = re.compile(r"scales_fusion_layers\.")


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)


def synth_param(name: str, shape, seed: int = 0) -> np.ndarray:
    """fp32 value for state-dict entry ``name`` of shape ``shape`` (int64 for counters)."""
    shape = tuple(int(s) for s in shape)
    rs = _rs(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 4:  # conv / deconv kernels
        fan_in = shape[1] * shape[2] * shape[3]
        if "deconv.0" in name:  # ConvTranspose2d weight is [Cin, Cout, kh, kw]; 4 of 16 taps hit a pixel
            fan_in = shape[0] * 4
        return (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    if len(shape) == 2:  # linear (ClassificationHRNet head)
        return (rs.standard_normal(shape) * np.sqrt(1.0 / shape[1])).astype(np.float32)
    if leaf == "weight":  # BN gamma
        if _LAST_BN.search(name):
            lo, hi = 0.1, 0.2
        elif _FUSION_BN.search(name):
            lo, hi = 0.15, 0.3
        else:
            lo, hi = 0.8, 1.2
        return rs.uniform(lo, hi, shape).astype(np.float32)
    if leaf == "bias":
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    raise ValueError(f"no synthetic rule for {name} {shape}")


def synth_state_dict(shapes: dict, seed: int = 0) -> dict:
    """``shapes``: {key: shape}. Returns {key: np.ndarray}."""
    return {k: synth_param(k, s, seed) for k, s in shapes.items()}


def synth_images(batch: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """Normalised-image-like input, NCHW fp32 (~N(0,1))."""
    rs = np.random.RandomState(1000003 + seed)
    return rs.standard_normal((batch, 3, h, w)).astype(np.float32)


# ----------------------------------------------------------------------------------------
# Constructed network-output maps for the decode path:
# P people, each a set of Gaussian blobs (sigma in quarter-res pixels) with a per-person
# tag constant + N(0, tag_noise); background U(0, bg).  Generated at the resolutions the
# net emits: heatmaps at 1/4 and 1/2, tags at 1/4 (one map per TTA pass).
# ----------------------------------------------------------------------------------------
def synth_decode_maps(
    num_kpts: int,
    hq: int,
    wq: int,
    num_people: int,
    seed: int = 0,
    emb: int = 1,
    sigma: float = 2.0,
    bg: float = 0.02,
    tag_noise: float = 0.05,
    drop_prob: float = 0.15,
    tag_spacing: float = 1.7,
):
    """Returns (hm_q [K,hq,wq], hm_h [K,2hq,2wq], tags [emb][K,hq,wq], people [P,K,3])."""
    rs = np.random.RandomState(424243 + seed)
    K = num_kpts
    hh, wh = 2 * hq, 2 * wq
    hm_q = rs.uniform(0.0, bg, (K, hq, wq)).astype(np.float32)
    hm_h = rs.uniform(0.0, bg, (K, hh, wh)).astype(np.float32)
    tags = [(tag_noise * rs.standard_normal((K, hq, wq))).astype(np.float32) for _ in range(emb)]
    people = np.zeros((num_people, K, 3), dtype=np.float32)
    yq, xq = np.mgrid[0:hq, 0:wq].astype(np.float32)
    yh, xh = np.mgrid[0:hh, 0:wh].astype(np.float32)
    for p in range(num_people):
        cx = rs.uniform(0.15 * wq, 0.85 * wq)
        cy = rs.uniform(0.15 * hq, 0.85 * hq)
        tagval = [tag_spacing * (p + 1) * (1 if e == 0 else -0.5) for e in range(emb)]
        for k in range(K):
            if rs.uniform() < drop_prob:
                continue
            x = float(np.clip(cx + rs.uniform(-0.12, 0.12) * wq, 2, wq - 3))
            y = float(np.clip(cy + rs.uniform(-0.12, 0.12) * hq, 2, hq - 3))
            amp = rs.uniform(0.5, 1.0)
            people[p, k] = (x, y, amp)
            gq = amp * np.exp(-((xq - x) ** 2 + (yq - y) ** 2) / (2 * sigma**2))
            gh = amp * np.exp(-((xh - 2 * x - 0.5) ** 2 + (yh - 2 * y - 0.5) ** 2) / (2 * (2 * sigma) ** 2))
            hm_q[k] = np.maximum(hm_q[k], gq.astype(np.float32))
            hm_h[k] = np.maximum(hm_h[k], gh.astype(np.float32))
            m = gq > 0.05 * amp
            for e in range(emb):
                tags[e][k][m] = (tagval[e] + tag_noise * rs.standard_normal(int(m.sum()))).astype(np.float32)
    return hm_q, hm_h, tags, people


# ----------------------------------------------------------------------------------------
# Training batch of BASELINE.json configs[2]: random people -> joints per stage, target heatmaps
# rendered by the restated HeatmapGenerator (sigma 2), masks.  Shapes follow keypoints/datasets/coco.py:140-164.
# ----------------------------------------------------------------------------------------
def synth_train_targets(batch: int, num_kpts: int, input_size: int, people, seed: int = 0, vis_prob: float = 0.8,
                        mask_holes: bool = False):
    """-> (heatmaps [2 x f32 [B,K,s,s]], masks [2 x f32 [B,s,s]], joints [2 x list of int32 [P_b,K,3]]) for the 1/4 and
    1/2 stages.  `people` = int or per-image list (0 allowed)."""
    from .keypoints.targets import HeatmapGenerator, JointsGenerator

    rs = np.random.RandomState(777001 + seed)
    counts = [people] * batch if np.isscalar(people) else list(people)
    sizes = [input_size // 4, input_size // 2]
    hms = [np.zeros((batch, num_kpts, s, s), np.float32) for s in sizes]
    masks = [np.ones((batch, s, s), np.float32) for s in sizes]
    joints = [[], []]
    for b in range(batch):
        P = counts[b]
        cx, cy = rs.uniform(0.1, 0.9, P) * input_size, rs.uniform(0.1, 0.9, P) * input_size
        raw = np.zeros((P, num_kpts, 3))
        for p in range(P):
            raw[p, :, 0] = cx[p] + rs.normal(0, input_size / 12, num_kpts)  # some fall outside the image on purpose
            raw[p, :, 1] = cy[p] + rs.normal(0, input_size / 12, num_kpts)
            raw[p, :, 2] = (rs.uniform(size=num_kpts) < vis_prob) * rs.randint(1, 3, num_kpts)
        for i, s in enumerate(sizes):
            sc = raw.copy()
            sc[..., :2] *= s / input_size
            j = JointsGenerator(s)(sc)
            joints[i].append(j)
            hms[i][b] = HeatmapGenerator(num_kpts, s, 2)(j)
            if mask_holes:  # crowd regions are masked out of the heatmap loss (coco.py:167-180)
                y0, x0 = rs.randint(0, s // 2, 2)
                masks[i][b, y0:y0 + s // 4, x0:x0 + s // 3] = 0
    return hms, masks, joints


def synth_train_preds(hms, seed: int = 0):
    """Predictions for a loss test: target heatmaps + N(0, 0.1) per stage, tags ~ N(0, 1) at the 1/4 stage."""
    rs = np.random.RandomState(9000 + seed)
    pred = [(h + rs.normal(0, 0.1, h.shape)).astype(np.float32) for h in hms]
    tags = rs.normal(0, 1.0, hms[0].shape).astype(np.float32)
    return pred, tags


def edit_loss_case(tag: str, joints):
    """Hand-made edge cases of the loss fixtures (shared by tools/make_golden.py and the tests)."""
    if tag == "b2_dups":  # two people sharing a pixel for one joint; people with no / a single visible joint
        joints[0][0][1, 5] = joints[0][0][0, 5] = (7, 9, 1)
        joints[0][0][2, :, :] = 0
        joints[0][1][1, :, 2] = 0
        joints[0][1][1, 3] = (4, 4, 1)
    return joints


# ----------------------------------------------------------------------------------------
# "Pass-through" HigherHRNet: seeded dense weights like synth_param, except that a few RESERVED channels carry values of the
# input image unchanged to the outputs, so that forward + decode can be exercised (and timed) as ONE chain on person-like
# maps without a trained checkpoint (none is available offline).  Every 4x4x3 block of the image holds, for its quarter-res
# pixel, the K heatmap values and one tag value; the two stem convs gather them into channels 0..K of the 64-channel
# tensor (one-hot taps), residual units are identities on the reserved channels (their conv rows are zero and every value
# is >= 0, so relu(0 + x) = x), the transition / fusion layers copy them, the 1x1 head reads them out (all K tag maps =
# the one tag channel) and the transposed conv carries the bilinear x2 kernel [.25 .75 .75 .25]^2 for the half-res
# heatmaps.  All other channels keep their random dense weights (they read the reserved channels but never write them),
# so the arithmetic the chip performs -- and the clock it holds -- is that of a dense net.
# ----------------------------------------------------------------------------------------
def synth_passthrough_state_dict(shapes: dict, num_kpts: int = 17, seed: int = 0, tag_gain: float = 1.0) -> dict:
    K, R = num_kpts, num_kpts + 1  # reserved channels 0..K-1 = heatmaps, K = tag field
    sd = {k: synth_param(k, s, seed).copy() for k, s in shapes.items()}

    def bn_identity(prefix, chans):
        for c in chans:
            sd[prefix + ".weight"][c] = 1.0
            sd[prefix + ".bias"][c] = 0.0
            sd[prefix + ".running_mean"][c] = 0.0
            sd[prefix + ".running_var"][c] = 1.0 - 1e-5

    def slot(v):  # value v of a quarter-res pixel sits at (dy, dx, ch) of its 4x4x3 image block
        return v // 12, (v // 3) % 4, v % 3

    # stem: image block -> 12 half-res channels ((dy&1, dx&1, ch)) -> R quarter-res channels
    w1, w2 = sd["backbone.conv1.weight"], sd["backbone.conv2.weight"]
    w1[:12] = 0.0
    for a in range(2):
        for b in range(2):
            for ch in range(3):
                w1[(a * 2 + b) * 3 + ch, ch, 1 + a, 1 + b] = 1.0
    bn_identity("backbone.bn1", range(12))
    w2[:R] = 0.0
    for v in range(R):
        dy, dx, ch = slot(v)
        w2[v, ((dy & 1) * 2 + (dx & 1)) * 3 + ch, 1 + (dy >> 1), 1 + (dx >> 1)] = 1.0
    bn_identity("backbone.bn2", range(R))
    for key in list(sd):
        w = sd[key]
        # residual units: the last conv of every unit writes nothing into the reserved channels
        if key.endswith((".conv2.weight", ".conv3.weight")) and ".scales_blocks." in key or ".resid_blocks." in key and key.endswith(".conv2.weight"):
            if ".stages.0." in key and key.endswith(".conv2.weight"):
                continue  # Bottleneck: conv2 is the middle conv, conv3 the last
            w[:R] = 0.0
            bn_identity(key.replace(".conv", ".bn").rsplit(".", 1)[0], range(R))
        # fusion terms into output scale 0 (1x1 conv + BN at low resolution, upsampled and added)
        if ".scales_fusion_layers.0." in key and key.endswith(".0.weight"):
            w[:R] = 0.0
            bn_identity(key.rsplit(".", 2)[0] + ".1", range(R))
    # stage 0, unit 0: the downsample branch carries the reserved channels from the 64- to the 256-channel tensor
    ds = "backbone.stages.0.blocks.0.scales_blocks.0.0.downsample"
    sd[ds + ".0.weight"][:R] = 0.0
    for v in range(R):
        sd[ds + ".0.weight"][v, v, 0, 0] = 1.0
    bn_identity(ds + ".1", range(R))
    # transition into branch 0: centre tap copy
    tr = "backbone.stages.0.transition_layer.transition_blocks.0"
    sd[tr + ".0.weight"][:R] = 0.0
    for v in range(R):
        sd[tr + ".0.weight"][v, v, 1, 1] = 1.0
    bn_identity(tr + ".1", range(R))
    # heads
    C = sd["init_heatmaps_head.weight"].shape[1]
    hw = sd["init_heatmaps_head.weight"]
    hw[:] = 0.0
    sd["init_heatmaps_head.bias"][:] = 0.0
    for k in range(K):
        hw[k, k, 0, 0] = 1.0       # heatmap k
        hw[K + k, K, 0, 0] = tag_gain   # tag map k = the tag channel (x tag_gain: uint8 images can only carry values up to ~2.2)
    dw = sd["deconv_layers.0.deconv.0.weight"]  # [C + 2K, C, 4, 4]
    dw[:, :K] = 0.0
    b = np.array([0.25, 0.75, 0.75, 0.25], np.float32)
    for k in range(K):
        dw[C + k, k] = np.outer(b, b)
    bn_identity("deconv_layers.0.deconv.1", range(K))
    fw = sd["deconv_layers.0.final_layer.weight"]
    fw[:] = 0.0
    sd["deconv_layers.0.final_layer.bias"][:] = 0.0
    for k in range(K):
        fw[k, k, 0, 0] = 1.0
    return sd


def synth_passthrough_images(batch: int, hq: int, wq: int, num_people, num_kpts: int = 17, seed: int = 0, tag_gain: float = 1.0,
                             sigma: float = 2.0):
    """Images [B,3,4hq,4wq] for a pass-through net: constructed quarter-res heatmaps (synth_decode_maps) and one tag field per
    image (per pixel the tag of the strongest blob there), rounded to bf16-exact values, written into the reserved slots of
    every 4x4x3 block; all other pixels ~ N(0,1).  Returns (images, hm_q [B,K,hq,wq], tag_field [B,hq,wq])."""
    K = num_kpts
    rs = np.random.RandomState(77 + seed)
    images = rs.standard_normal((batch, 3, 4 * hq, 4 * wq)).astype(np.float32)
    hms = np.zeros((batch, K, hq, wq), np.float32)
    fields = np.zeros((batch, hq, wq), np.float32)
    yq, xq = np.mgrid[0:hq, 0:wq].astype(np.float32)

    def bf16(a):  # round to nearest even to bf16, back to fp32
        u = a.astype(np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).view(np.float32)

    for i in range(batch):
        P = num_people[i % len(num_people)] if isinstance(num_people, (list, tuple)) else num_people
        hm_q, _, _, people = synth_decode_maps(K, hq, wq, P, seed=seed * 1000 + i, sigma=sigma)
        field = np.abs(0.05 * rs.standard_normal((hq, wq))).astype(np.float32)
        best = np.zeros((hq, wq), np.float32)
        for p in range(P):
            for k in range(K):
                x, y, amp = people[p, k]
                if amp == 0:
                    continue
                g = amp * np.exp(-((xq - x) ** 2 + (yq - y) ** 2) / (2 * 2.0**2))
                m = (g > 0.05 * amp) & (g > best)
                field[m] = 1.7 * (p + 1) + 0.05 * rs.standard_normal(int(m.sum()))
                best = np.maximum(best, g)
        hms[i], fields[i] = bf16(hm_q), bf16(field)
        vals = np.concatenate([hms[i], fields[i][None] / tag_gain], 0)  # [K+1,hq,wq]
        for v in range(K + 1):
            dy, dx, ch = v // 12, (v // 3) % 4, v % 3
            images[i, ch, dy::4, dx::4] = vals[v]
    return images, hms, fields


def synth_passthrough_raw_u8(batch: int, hq: int, wq: int, num_people, num_kpts: int = 17, seed: int = 0, tag_gain: float = 8.0):
    """The same as raw uint8 HWC images for InferenceKeypointsModel (ToTensor + ImageNet normalisation undone and rounded to
    pixel values: the encoded values are quantised to ~0.017, the tag channel is stored / tag_gain to fit the pixel range)."""
    mean = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
    x = synth_passthrough_images(batch, hq, wq, num_people, num_kpts, seed, tag_gain)[0]
    px = np.clip(np.rint((x * std + mean) * 255.0), 0, 255).astype(np.uint8)
    return [np.ascontiguousarray(px[i].transpose(1, 2, 0)) for i in range(batch)]
