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
