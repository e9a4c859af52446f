#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference and the pinned munkres 1.1.4 that
ships in the container's conda tree); nothing here travels to the GPU box except the .npz /
.json files it writes.  Recipe from SURVEY.md §8c:

  PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden.py

What is imported from the reference (and therefore *pinned* by these fixtures):
  * src.keypoints.architectures.higher_hrnet.HigherHRNet         (net forward, a1-a9)
  * src.classification.architectures.hrnet.ClassificationHRNet   (cfg 1, a22)
  * src.keypoints.grouping.MPPEHeatmapParser                     (decode, a12-a17)
  * munkres.Munkres 1.1.4                                        (assignment, a14)
  * src.base.transforms.utils.get_multi_scale_size               (resize geometry, a10)
  * src.keypoints.loss.AEKeypointsLoss                           (training loss + autograd gradients, a20)
  * the same HigherHRNet in .train() mode + torch autograd        (train-mode forward / backward, a20)
What is NOT importable here (cv2 / torchvision missing) and is restated inline with the
same torch calls the reference makes: the three F.interpolate(bilinear,
align_corners=False) + stack/mean lines of results.py:48-67,225-230 and the flip-TTA
lines of model.py:85-94.

Inputs and weights are produced by this repo's own seeded generators (synth.py), so the
fixtures hold outputs only.
"""
import hashlib
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MUNKRES = "/opt/conda/lib/python3.9/site-packages/munkres.py"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

spec = importlib.util.spec_from_file_location("munkres", MUNKRES)
munkres = importlib.util.module_from_spec(spec)
spec.loader.exec_module(munkres)
assert munkres.__version__ == "1.1.4"
sys.modules["munkres"] = munkres

synth = importlib.import_module("pytorch-human-pose_amd.synth")
from src.classification.architectures.hrnet import ClassificationHRNet  # noqa: E402
from src.keypoints.architectures.higher_hrnet import HigherHRNet  # noqa: E402
from src.keypoints.grouping import MPPEHeatmapParser  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

COCO_FLIP_INDEX = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]  # transforms.py:11


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_synth(net: torch.nn.Module, seed: int) -> None:
    sd = net.state_dict()
    new = {k: torch.from_numpy(synth.synth_param(k, v.shape, seed)) for k, v in sd.items()}
    net.load_state_dict(new, strict=True)
    net.eval()


# ------------------------------------------------------------------ net forward fixtures
def net_fixtures():
    out = {}
    # (tag, C, B, H, W, seed, taps?)
    cases = [("w32_64", 32, 1, 64, 64, 0, True), ("w32_128", 32, 2, 128, 128, 1, False),
             ("w32_96x160", 32, 1, 96, 160, 2, False), ("w48_64", 48, 1, 64, 64, 3, False)]
    for tag, C, B, H, W, seed, taps in cases:
        net = HigherHRNet(17, C)
        load_synth(net, seed)
        x = torch.from_numpy(synth.synth_images(B, H, W, seed))
        captured = {}
        hooks = []
        if taps:
            def mk(name):
                def hook(_m, _inp, outp):
                    ts = outp if isinstance(outp, (list, tuple)) else [outp]
                    for i, t in enumerate(ts):
                        captured[f"{name}#{i}"] = t.detach().clone().numpy()
                return hook
            hooks.append(net.backbone.stages[0].register_forward_pre_hook(
                lambda _m, inp: captured.__setitem__("stem#0", inp[0].detach().clone().numpy())))
            for s, st in enumerate(net.backbone.stages):
                for b, blk in enumerate(st.blocks):
                    hooks.append(blk.register_forward_hook(mk(f"stages.{s}.blocks.{b}")))
                hooks.append(st.register_forward_hook(mk(f"stages.{s}")))
            hooks.append(net.deconv_layers[0].register_forward_hook(mk("deconv")))
        with torch.no_grad():
            hms, tags = net(x)
        for h in hooks:
            h.remove()
        out[f"{tag}/hm_q"] = hms[0].numpy()
        out[f"{tag}/hm_h"] = hms[1].numpy()
        out[f"{tag}/tags"] = tags.numpy()
        for k, v in captured.items():
            out[f"{tag}/tap/{k}"] = v
        print(tag, [tuple(h.shape) for h in hms], float(hms[0].abs().max()), float(hms[1].abs().max()),
              float(tags.abs().max()), len(captured))
    # full-size W32 512x512: sampled positions + checksums only
    net = HigherHRNet(17, 32)
    load_synth(net, 0)
    x = torch.from_numpy(synth.synth_images(1, 512, 512, 7))
    with torch.no_grad():
        hms, tags = net(x)
    rs = np.random.RandomState(5)
    for name, t in (("hm_q", hms[0]), ("hm_h", hms[1]), ("tags", tags)):
        a = t.numpy()
        idx = rs.randint(0, a.size, 4096)
        out[f"w32_512/{name}_idx"] = idx.astype(np.int64)
        out[f"w32_512/{name}_val"] = a.reshape(-1)[idx]
        out[f"w32_512/{name}_stats"] = np.array([a.mean(), a.std(), np.abs(a).max()], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "net_forward.npz"), **out)

    # cfg 1: ClassificationHRNet-W32 on one 224x224 image
    cnet = ClassificationHRNet(32, 1000)
    load_synth(cnet, 11)
    x = torch.from_numpy(synth.synth_images(1, 224, 224, 11))
    with torch.no_grad():
        logits = cnet(x)
    np.savez_compressed(os.path.join(OUT, "cls_forward.npz"), logits=logits.numpy())
    print("cls logits", float(logits.abs().max()))


# ------------------------------------------------------------------ decode fixtures
def aggregate(hm_q, hm_h, tags_list):
    """results.py:225-234 restated with the same torch calls (module not importable: cv2)."""
    hq = torch.from_numpy(hm_q)[None]
    hh = torch.from_numpy(hm_h)[None]
    H, W = hh.shape[-2] * 2, hh.shape[-1] * 2
    interp = torch.nn.functional.interpolate
    up = interp(hq, size=list(hh.shape[-2:]), mode="bilinear", align_corners=False)
    avg = torch.stack([up, hh]).mean(dim=0)
    full = interp(avg, size=[H, W], mode="bilinear", align_corners=False)
    tfull = [interp(torch.from_numpy(t)[None], size=[H, W], mode="bilinear", align_corners=False) for t in tags_list]
    tfull = torch.stack(tfull, dim=4)
    return full[0], tfull[0]


DECODE_CASES = [
    # tag, hq, wq, people, seed, emb, det_thr, tag_thr, max_people, kwargs
    # NB: torch's CPU bilinear switches to a differently-rounded vectorised kernel when
    # out_h + out_w <= 128 (ATen UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d); real inputs
    # (>= 256 px) never reach it, so every case keeps 2*hq + 2*wq > 128.
    ("p3_160", 40, 40, 3, 1, 1, 0.05, 0.5, 30, {}),
    ("p3_160_e2", 40, 40, 3, 2, 2, 0.05, 0.5, 30, {}),
    ("p0_160", 40, 40, 0, 3, 1, 0.05, 0.5, 30, {}),
    ("p1_160", 40, 40, 1, 4, 1, 0.05, 0.5, 30, {}),
    ("p5_ragged", 32, 48, 5, 5, 1, 0.05, 0.5, 30, {}),
    ("p5_ragged_b", 56, 40, 5, 15, 1, 0.05, 0.5, 30, {}),
    ("p6_missing", 48, 48, 6, 6, 1, 0.05, 0.5, 30, {"drop_prob": 0.5}),
    ("p4_zero_bg", 40, 40, 4, 7, 1, 0.05, 0.5, 30, {"bg": 0.0}),
    ("p8_val_thr", 48, 48, 8, 8, 1, 0.1, 1.0, 20, {}),
    ("p10_512", 128, 128, 10, 9, 1, 0.05, 0.5, 30, {}),
    ("p10_512_e2", 128, 128, 10, 10, 2, 0.05, 0.5, 30, {}),
    ("p35_512", 128, 128, 35, 11, 1, 0.05, 0.5, 30, {"sigma": 1.5}),
    ("p12_close_tags", 64, 64, 12, 12, 1, 0.05, 0.5, 30, {"tag_spacing": 0.4, "tag_noise": 0.15}),
    ("p40_cap5", 64, 64, 40, 13, 1, 0.05, 0.5, 5, {"sigma": 1.5}),
]


def decode_fixtures():
    out = {}
    meta = {}
    for tag, hq, wq, P, seed, emb, det, tthr, maxp, kw in DECODE_CASES:
        hm_q, hm_h, tags, _ = synth.synth_decode_maps(17, hq, wq, P, seed=seed, emb=emb, **kw)
        full, tfull = aggregate(hm_q, hm_h, tags)
        parser = MPPEHeatmapParser(17, max_num_people=maxp, det_thr=det, tag_thr=tthr)
        tags_k, coords_k, scores_k = parser.top_k(full, tfull)
        grouped = parser.match_by_tag(tags_k, coords_k, scores_k)
        if len(grouped):
            adjusted = parser.adjust(grouped.copy(), full.numpy())
        else:
            adjusted = grouped
        joints, scores = parser.parse(full, tfull, adjust=True, refine=True)
        joints_nr, scores_nr = parser.parse(full, tfull, adjust=True, refine=False)
        joints_na, _ = parser.parse(full, tfull, adjust=False, refine=True)
        out[f"{tag}/tags_k"] = tags_k
        out[f"{tag}/coords_k"] = coords_k
        out[f"{tag}/scores_k"] = scores_k
        out[f"{tag}/grouped"] = grouped
        out[f"{tag}/adjusted"] = adjusted
        out[f"{tag}/joints"] = joints
        out[f"{tag}/scores"] = scores
        out[f"{tag}/joints_norefine"] = joints_nr
        out[f"{tag}/joints_noadjust"] = joints_na
        meta[tag] = dict(hq=hq, wq=wq, people=P, seed=seed, emb=emb, det_thr=det, tag_thr=tthr, max_people=maxp,
                         kwargs=kw, full_hm_sha256=sha(full.numpy()), full_tags_sha256=sha(tfull.numpy()),
                         num_found=int(len(joints)),
                         # exact ties among candidates: torch.topk's order between equal values is
                         # unspecified (CPU heap sort vs CUDA radix select), tests compare those as sets
                         has_ties=bool(any(len(np.unique(r[r > 0])) != int((r > 0).sum()) for r in scores_k)))
        print(tag, "found", len(joints), "of", P, "topk min score", float(scores_k.min()), "ties", meta[tag]["has_ties"])
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **out)
    with open(os.path.join(OUT, "decode_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


# ------------------------------------------------------------------ flip TTA fixture
def flip_fixture():
    """model.py:85-94 restated (module not importable: torchvision). Net is the reference's."""
    net = HigherHRNet(17, 32)
    load_synth(net, 0)
    x = torch.from_numpy(synth.synth_images(1, 64, 64, 21))
    with torch.no_grad():
        hms, tags = net(x)
        fh, ft = net(torch.flip(x, [3]))
        hms = [(hms[i] + torch.flip(fh[i], [3])[:, COCO_FLIP_INDEX]) / 2 for i in range(2)]
        tags2 = torch.flip(ft, [3])[:, COCO_FLIP_INDEX]
    np.savez_compressed(os.path.join(OUT, "flip_tta.npz"), hm_q=hms[0].numpy(), hm_h=hms[1].numpy(),
                        tags0=tags.numpy(), tags1=tags2.numpy())


# ------------------------------------------------------------------ munkres fixtures
def munkres_fixtures():
    rs = np.random.RandomState(77)
    mats, res = {}, {}
    for i in range(60):
        r = int(rs.randint(1, 31))
        c = int(rs.randint(1, 31))
        kind = i % 4
        if kind == 0:
            m = rs.uniform(0, 10, (r, c))
        elif kind == 1:  # the reference's cost shape: round(dist)*100 - score, many ties
            m = np.round(rs.uniform(0, 3, (r, c))) * 100 - rs.uniform(0.05, 1.0, (r, 1)).astype(np.float32).astype(np.float64)
        elif kind == 2:  # with the 1e10 padding columns of grouping.py:126-128
            m = np.round(rs.uniform(0, 2, (r, c))) * 100 - rs.uniform(0.05, 1.0, (r, 1))
            if r > c:
                m = np.concatenate([m, np.zeros((r, r - c)) + 1e10], axis=1)
        else:  # small integers, heavy ties
            m = rs.randint(0, 4, (r, c)).astype(np.float64)
        m = np.ascontiguousarray(m, dtype=np.float64)
        # numpy input as grouping.py:55-59 passes it (needs cols >= rows: munkres' pad_matrix cannot
        # extend numpy rows); otherwise plain lists, the library's documented input type.
        pairs = munkres.Munkres().compute(m.copy() if m.shape[1] >= m.shape[0] else m.tolist())
        mats[f"m{i}"] = m
        res[f"r{i}"] = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    np.savez_compressed(os.path.join(OUT, "munkres.npz"), **mats, **res)
    print("munkres cases", len(mats))


# ------------------------------------------------------------------ resize geometry
def geometry_fixtures():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # utils.py:2 imports cv2; the function below never touches it
    from src.base.transforms.utils import get_multi_scale_size
    rows = []
    for (h, w) in [(480, 640), (640, 480), (427, 640), (512, 512), (333, 500), (1080, 1920), (375, 500)]:
        for input_size in (512, 640):
            for cur, mn in [(1, 1), (0.5, 0.5), (1, 0.5), (2, 0.5)]:
                size, center, scale = get_multi_scale_size(np.zeros((h, w, 3), np.uint8), input_size, cur, mn)
                rows.append(dict(h=h, w=w, input_size=input_size, current_scale=cur, min_scale=mn,
                                 size=[int(size[0]), int(size[1])], center=[int(center[0]), int(center[1])],
                                 scale=[float(scale[0]), float(scale[1])]))
    with open(os.path.join(OUT, "multi_scale_size.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print("geometry rows", len(rows))


# (tag, B, input_size, people per image, seed, mask holes, hand-made edits)
LOSS_CASES = [
    ("b3_128", 3, 128, [3, 1, 0], 1, True),
    ("b4_256", 4, 256, [10, 7, 2, 5], 2, True),
    ("b2_dups", 2, 128, [4, 2], 3, False),
    ("b2_empty", 2, 128, [0, 0], 4, False),
]


def loss_fixtures():
    """AEKeypointsLoss.calculate_loss (loss.py:64-93) + the sum of module.py:50-59, fp32 on CPU, and torch autograd's
    gradients of that sum w.r.t. the two predicted heatmap stacks and the predicted tag map."""
    from src.keypoints.loss import AEKeypointsLoss

    out = {}
    for tag, B, size, people, seed, holes in LOSS_CASES:
        K = 17
        hms, masks, joints = synth.synth_train_targets(B, K, size, people, seed=seed, mask_holes=holes)
        joints = synth.edit_loss_case(tag, joints)
        p_np, t_np = synth.synth_train_preds(hms, seed)
        pred = [torch.from_numpy(p).requires_grad_() for p in p_np]
        tags = torch.from_numpy(t_np).requires_grad_()
        loss_fn = AEKeypointsLoss()
        hl, push, pull = loss_fn.calculate_loss(pred, tags, [torch.from_numpy(h) for h in hms], [torch.from_numpy(m) for m in masks], joints)
        total = hl[0] + hl[1] + push[0] + pull[0]
        total = total if torch.is_tensor(total) else torch.tensor(float(total))
        if total.requires_grad:
            total.backward()
        z = lambda t: np.zeros(tuple(t.shape), np.float32) if t.grad is None else t.grad.numpy()
        f = lambda v: np.float32(v.item() if torch.is_tensor(v) else v)
        out[f"{tag}.losses"] = np.array([f(hl[0]), f(hl[1]), f(push[0]), f(pull[0]), f(total)], np.float32)
        # dense heatmap gradients: float64 sums + 512 sampled entries per stage; the tag gradient is sparse: store it whole
        for i in range(2):
            g = z(pred[i]).ravel()
            idx = np.random.RandomState(5 + i).randint(0, g.size, 512)
            out[f"{tag}.g_pred{i}_idx"], out[f"{tag}.g_pred{i}_val"] = idx.astype(np.int64), g[idx]
            out[f"{tag}.g_pred{i}_sums"] = np.array([g.astype(np.float64).sum(), np.abs(g.astype(np.float64)).sum()])
        gt = z(tags).ravel()
        nz = np.flatnonzero(gt)
        out[f"{tag}.g_tags_idx"], out[f"{tag}.g_tags_val"] = nz.astype(np.int64), gt[nz]
        print(tag, out[f"{tag}.losses"], "nonzero tag grads", nz.size)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)
    json.dump({"cases": [list(c) for c in LOSS_CASES]}, open(os.path.join(OUT, "loss_meta.json"), "w"), indent=1)


def train_fixture(batch=2, fname="train_step.npz"):
    """The reference HigherHRNet in .train() mode (batch-statistics BatchNorm): forward on a seeded batch, the scalar
    mean(hm0^2) + mean(hm1^2) + mean(tags^2), torch autograd gradients of it, and the running statistics after the step.
    batch = 8 (train_step_b8.npz): the 256-channel branch then normalises over 128 samples per channel instead of 32, so the
    batch statistics themselves no longer move with bf16 rounding and the end-to-end comparison can be tight."""
    net = HigherHRNet(17, 32)
    load_synth(net, 5)
    net.train()
    x = torch.from_numpy(synth.synth_images(batch, 128, 128, seed=1))
    hms, tags = net(x)
    loss = (hms[0] ** 2).mean() + (hms[1] ** 2).mean() + (tags ** 2).mean()
    loss.backward()
    out = {"loss": np.float32(loss.item())}
    rs = np.random.RandomState(3)
    for name, t in (("hm0", hms[0]), ("hm1", hms[1]), ("tags", tags)):
        a = t.detach().numpy().ravel()
        idx = rs.randint(0, a.size, 256)
        out[f"{name}.idx"], out[f"{name}.val"], out[f"{name}.absmax"] = idx.astype(np.int64), a[idx], np.float32(np.abs(a).max())
    names, norms, samples = [], [], []
    for name, p in net.named_parameters():
        g = p.grad.numpy().ravel()
        names.append(name)
        norms.append(np.linalg.norm(g.astype(np.float64)))
        samples.append(g[np.linspace(0, g.size - 1, 4).astype(int)])
    out["grad.names"] = np.array(names)
    out["grad.norms"] = np.array(norms, np.float64)
    out["grad.samples"] = np.stack(samples).astype(np.float32)
    sd = net.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.bn1.running_var", "deconv_layers.0.deconv.1.running_mean",
              "backbone.stages.3.blocks.4.scales_blocks.3.3.bn2.running_var"):
        out["stat." + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print("train fixture", fname, ": loss", out["loss"], "params", len(names))


def train_autocast_fixture():
    """The same step at the REFERENCE'S training precision (keypoints/module.py:48-60: forward under
    torch.autocast(dtype=float16), backward through a GradScaler): CPU autocast runs the reference net with fp16 conv operands
    / fp16 activations here.  The loss scale is what GradScaler's halving converges to from its initial 65536: the largest
    power of two that leaves every gradient finite.  Stored like train_step.npz (same sample indices), so that the bf16 engine's
    deviation from the fp32 step can be put beside the reference's own mixed-precision deviation."""
    x = torch.from_numpy(synth.synth_images(2, 128, 128, seed=1))
    scale = 65536.0
    while True:
        net = HigherHRNet(17, 32)
        load_synth(net, 5)
        net.train()
        with torch.autocast(device_type="cpu", dtype=torch.float16):
            hms, tags = net(x)
            loss = (hms[0].float() ** 2).mean() + (hms[1].float() ** 2).mean() + (tags.float() ** 2).mean()
        (loss * scale).backward()
        if all(torch.isfinite(p.grad).all() for p in net.parameters()):
            break
        scale /= 2  # GradScaler: skip the step, halve the scale
    out = {"loss": np.float32(loss.item()), "loss_scale": np.float64(scale)}
    rs = np.random.RandomState(3)
    for name, t in (("hm0", hms[0]), ("hm1", hms[1]), ("tags", tags)):
        a = t.detach().float().numpy().ravel()
        idx = rs.randint(0, a.size, 256)
        out[f"{name}.idx"], out[f"{name}.val"] = idx.astype(np.int64), a[idx]
    names, norms, samples = [], [], []
    for name, p in net.named_parameters():
        g = (p.grad / scale).numpy().ravel()
        names.append(name)
        norms.append(np.linalg.norm(g.astype(np.float64)))
        samples.append(g[np.linspace(0, g.size - 1, 4).astype(int)])
    out["grad.names"] = np.array(names)
    out["grad.norms"] = np.array(norms, np.float64)
    out["grad.samples"] = np.stack(samples).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "train_step_autocast.npz"), **out)
    print("autocast train fixture: loss", out["loss"], "scale", scale)


# ------------------------------------------------------------------ COCO result packing (a19)
def eval_packing_fixture():
    """src/keypoints/bin/eval.py:18-49 `evaluate_dataset`, the reference's own function, run on a fake model and a fake dataset.
    Its module imports third-party packages that are absent here (pycocotools, cv2, torchvision, albumentations, mlflow, ...):
    none of them is touched by `evaluate_dataset`, so empty stand-in MODULE OBJECTS (attribute access yields a dummy class) satisfy
    the import lines; the code that runs and is pinned is the reference's packing loop, byte for byte."""

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Dummy()

        def __getattr__(self, name):
            return _Dummy()

        def __mro_entries__(self, bases):
            return (object,)

        def __iter__(self):
            return iter(())

        def __getitem__(self, k):  # typing-style subscripts (Generic[T], Callable[..., X])
            return _Dummy()

        def __or__(self, other):
            return _Dummy()

        __ror__ = __or__

    class _Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Dummy()

    import importlib.abc
    import importlib.machinery

    absent = ("pycocotools", "cv2", "torchvision", "albumentations", "mlflow", "colorlog", "natsort", "seaborn", "pynvml", "torchinfo",
              "thop", "dacite", "geda", "onnx", "onnxruntime", "plotly", "matplotlib", "PIL", "imageio", "moviepy", "pandas", "scipy",
              "skimage", "sklearn", "yaml", "psutil", "GPUtil", "rich", "dotenv", "git", "joblib", "gdown", "ffmpeg", "kaleido")

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path=None, target=None):
            top = name.split(".")[0]
            if top in absent:
                try:  # installed for real? then use it
                    for f in sys.meta_path:
                        if f is self:
                            continue
                        sp = f.find_spec(name, path, target) if hasattr(f, "find_spec") else None
                        if sp is not None and sp.origin not in (None, "namespace"):  # (a bare directory of that name is not the package)
                            return sp
                except Exception:  # noqa: BLE001
                    pass
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Stub(spec.name)

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _Finder())
    from src.keypoints.bin.eval import evaluate_dataset

    rs = np.random.RandomState(11)
    people = [3, 0, 1, 30]
    stems = ["000000000139", "000000397133", "000000000785", "100000000001"]
    cases = []
    for n in people:
        coords = (rs.uniform(-20, 700, (n, 17, 2))).astype(np.float32)   # what transform_coords leaves: float32 raw-image pixels
        scores = rs.uniform(0.01, 1.0, (n,)).astype(np.float32)           # person scores (grouping.py:276), float32
        cases.append((coords, scores))

    class Result:
        def __init__(self, c, s):
            self.kpts_coords, self.obj_scores = c, s

    class Model:
        def __init__(self):
            self.i = 0

        def __call__(self, raw_image, annot=None):
            assert annot is None and raw_image.shape == (8, 8, 3)
            r = Result(*cases[self.i])
            self.i += 1
            return r

    class Dataset:
        images_filepaths = [f"/data/COCO/images/val2017/{s}.jpg" for s in stems]

        def __len__(self):
            return len(stems)

        def load_image(self, idx):
            return np.zeros((8, 8, 3), np.uint8)

    results = evaluate_dataset(Model(), Dataset())
    assert len(results) == sum(people)
    np.savez_compressed(os.path.join(OUT, "eval_packing_inputs.npz"), stems=np.array(stems),
                        **{f"coords{i}": c for i, (c, _) in enumerate(cases)}, **{f"scores{i}": s for i, (_, s) in enumerate(cases)})
    with open(os.path.join(OUT, "eval_packing.json"), "w") as f:
        json.dump(results, f)
    print("eval packing fixture:", len(results), "entries; first", {k: (v if k != "keypoints" else v[:6]) for k, v in results[0].items()})


if __name__ == "__main__":
    if sys.argv[1:] == ["eval_packing"]:
        eval_packing_fixture()
        sys.exit(0)
    which = sys.argv[1:] or ["net", "decode", "flip", "munkres", "geometry", "loss", "train", "train_autocast"]
    if "train_autocast" in which:
        train_autocast_fixture()
    if "train" in which:
        train_fixture()
    if "train_b8" in which:
        train_fixture(8, "train_step_b8.npz")
    if "loss" in which:
        loss_fixtures()
    if "munkres" in which:
        munkres_fixtures()
    if "geometry" in which:
        geometry_fixtures()
    if "decode" in which:
        decode_fixtures()
    if "flip" in which:
        flip_fixture()
    if "net" in which:
        net_fixtures()
