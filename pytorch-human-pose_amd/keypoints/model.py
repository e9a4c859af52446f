"""Inference wrapper with the reference's interface.

Stands in for `/root/reference/src/keypoints/model.py:43-111` (`InferenceKeypointsModel`) and
`/root/reference/src/base/model.py:155-175` (checkpoint loading: weights under
ckpt["module"]["model"] or a bare state dict, prefixes `module.` / `_orig_mod.` / `net.` stripped).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor, nn

from .. import _lib
from .grouping import MPPEHeatmapParser
from .results import InferenceKeypointsResult, transform_coords
import ctypes as C

from .transforms_utils import COCO_FLIP_INDEX, IMAGENET_MEAN, IMAGENET_STD, dst_to_src_matrix, get_multi_scale_size

COCO_LIMBS = [(15, 13), (13, 11), (16, 14), (14, 12), (11, 12), (5, 11), (6, 12), (5, 6), (5, 7), (6, 8), (7, 9), (8, 10),
              (1, 2), (0, 1), (0, 2), (1, 3), (2, 4), (3, 5), (4, 6)]


def parse_checkpoint(ckpt: dict) -> dict:
    """utils/model.py:166-173"""
    out = {}
    for k, v in ckpt.items():
        for p in ("module.", "_orig_mod.", "net."):
            k = k.replace(p, "")
        out[k] = v
    return out


class KeypointsModel:
    """Training-side model wrapper: `KeypointsModel` (keypoints/model.py:15-40) on `BaseModel` (base/model.py:15-131)
    without the export / summary helpers (onnx, torchinfo: out of scope)."""

    def __init__(self, net: nn.Module):
        self.net = net
        self.input_names, self.output_names = ["images"], ["keypoints"]

    def _bare(self) -> nn.Module:
        from torch.nn.parallel import DistributedDataParallel as DDP
        return self.net.module if isinstance(self.net, DDP) else self.net

    def forward(self, images: Tensor):
        return self.net(images)

    def init_weights(self) -> None:
        """keypoints/model.py:19-34: conv / transposed-conv weights ~ N(0, 0.001), their biases 0, BatchNorm weight 1 / bias 0."""
        net = self._bare()
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        net.mark_dirty()

    def init_pretrained_weights(self, ckpt: dict) -> None:
        """base/model.py:101-123: load the checkpoint entries whose names exist here, ignore the rest."""
        net = self._bare()
        names = {n for n, _ in net.named_parameters()} | {n for n, _ in net.named_buffers()}
        net.load_state_dict({k: v for k, v in parse_checkpoint(ckpt).items() if k in names}, strict=False)

    def to_CUDA(self, device_id: int) -> None:
        self.net = self.net.cuda(device_id)

    def to_DDP(self, device_id: int, use_batchnorm: bool, bf16_gradients: bool = False) -> None:
        """base/model.py:36-48.  `use_batchnorm` is the reference's SyncBatchNorm switch (its trainer's default; the HigherHRNet
        experiment file sets `sync_batchnorm: false`): here the
        BatchNorm leaves stay what they are and the training forward shares their statistics across the ranks of the default
        process group (hh_bn_train_stats -> all-reduce -> hh_bn_train_normalize).  Gradients: torch DDP's bucketed
        all-reduce on RCCL, overlapped with the backward; `bf16_gradients=True` sends the buckets as bf16."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        self.net.sync_batchnorm = True if use_batchnorm else None
        self.net = DDP(self.net, device_ids=[device_id])
        if bf16_gradients:  # not in the reference: halves the 114.6 MB (W32) the ring moves per step; the optimizer still sees fp32
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            self.net.register_comm_hook(None, default_hooks.bf16_compress_hook)

    @property
    def device(self):
        return next(self._bare().parameters()).device

    def freeze(self) -> None:
        for p in self.net.parameters():
            p.requires_grad = False

    def state_dict(self) -> dict:
        return self._bare().state_dict()

    def load_state_dict(self, state_dict: dict) -> None:
        self._bare().load_state_dict(state_dict)

    def example_input(self) -> dict[str, Tensor]:
        return {"images": torch.randn(1, 3, 512, 512, device=self.device)}


class KeypointsModule:
    """`KeypointsModule.training_step` (keypoints/module.py:43-71): forward, AE loss, backward, optimizer step.  The reference
    runs under fp16 autocast with a GradScaler; this path keeps activations in bf16 (no scaler needed) and parameters fp32."""

    def __init__(self, model: KeypointsModel, loss_fn, optimizer: torch.optim.Optimizer):
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer

    def batch_to_device(self, batch):
        images, heatmaps, masks, joints = batch
        dev = self.model.device
        return images.to(dev), [h.to(dev, non_blocking=True) for h in heatmaps], [m.to(dev, non_blocking=True) for m in masks], joints

    def training_step(self, batch, batch_idx: int = 0) -> dict[str, float]:
        images, heatmaps, masks, joints = batch
        stages_hms, tags = self.model.net(images)
        hm_losses, push_losses, pull_losses = self.loss_fn.calculate_loss(stages_hms, tags, heatmaps, masks, joints)
        loss = 0
        for hl in hm_losses:
            loss = loss + hl
        loss = loss + push_losses[0] + pull_losses[0]
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.model._bare().mark_dirty()
        metrics = {"loss": loss.detach().item()}
        for i, hl in enumerate(hm_losses):
            metrics[f"hm_{i}_loss"] = hl.item()
        for i in range(len(push_losses)):
            metrics[f"push_{i}_loss"] = push_losses[i].item()
            metrics[f"pull_{i}_loss"] = pull_losses[i].item()
        return metrics

    def validation_step(self, batch, batch_idx: int = 0):
        """`KeypointsModule.validation_step` (keypoints/module.py:73-111): forward, the same losses as the training step (no
        backward), and one KeypointsResult per image decoded at the validation thresholds (20 people, det 0.1, tag 1.0).  The
        reference decodes image by image on the host; here the whole batch goes through one hh_decode."""
        from .results import KeypointsResult
        images, heatmaps, masks, joints = batch
        with torch.no_grad():
            stages_hms, tags = self.model.net(images)
            hm_losses, push_losses, pull_losses = self.loss_fn.calculate_loss(stages_hms, tags, heatmaps, masks, joints)
        loss = hm_losses[0] + hm_losses[1] + push_losses[0] + pull_losses[0]
        metrics = {"loss": loss.detach().item()}
        for i, hl in enumerate(hm_losses):
            metrics[f"hm_{i}_loss"] = hl.item()
        for i in range(len(push_losses)):
            metrics[f"push_{i}_loss"] = push_losses[i].item()
            metrics[f"pull_{i}_loss"] = pull_losses[i].item()
        stages_hms, tags = [h.detach().float() for h in stages_hms], tags.detach().float()
        # one parser (= one device workspace) for the life of the module, not one per step
        K = stages_hms[0].shape[1]
        parser = getattr(self, "_val_parser", None)
        if parser is None or parser.num_kpts != K:
            parser = self._val_parser = MPPEHeatmapParser(K, 20, 0.1, 1.0)
        cpu_images = images.detach().cpu()
        results = [KeypointsResult(cpu_images[i], [h[i:i + 1] for h in stages_hms], tags[i:i + 1], COCO_LIMBS, 20, 0.1, 1.0, parser=parser)
                   for i in range(len(cpu_images))]
        KeypointsResult.set_preds_batch(results, stages_hms, tags)
        return metrics, results


# hh_image_desc of include/hhrnet.h (64 bytes): byte offset of the image in the batch buffer, its size, destination -> source affine
_IMAGE_DESC = np.dtype([("offset", "<i8"), ("h", "<i4"), ("w", "<i4"), ("inv", "<f8", (6,))])


class InferenceKeypointsModel:
    limbs = COCO_LIMBS

    def __init__(self, net: nn.Module, det_thr: float = 0.05, tag_thr: float = 0.5, use_flip: bool = False,
                 input_size: int = 512, max_num_people: int = 30, device: str = "cuda:0", ckpt_path: str | None = None):
        self.net = net.to(device)
        self.net.eval()
        self.device = device
        self.input_size = input_size
        self.det_thr, self.tag_thr = det_thr, tag_thr
        self.max_num_people = max_num_people
        self.use_flip = use_flip
        self._lib = _lib.load()
        self._parser = MPPEHeatmapParser(net.num_kpts, max_num_people, det_thr, tag_thr)
        self._perm = np.asarray(COCO_FLIP_INDEX, np.int32)
        self._stream = None       # highest-priority compute stream (created on first use: needs the device)
        self._copy_stream = None  # infer_images: host -> device of the next batch's pixels
        self._d2h_stream = None   # infer_images: result copies, beside the next batch's launches
        self._out_ring: dict = {}  # infer_images: pinned result buffers per pipeline slot and shape
        if ckpt_path is not None:
            self.load_checkpoint(ckpt_path)

    def load_checkpoint(self, ckpt_path: str) -> None:
        """base/model.py:167-175: a trainer checkpoint keeps the weights under ["module"]["model"]; a bare state dict (the
        published pretrained/higher_hrnet_32.pt) is loaded as it is."""
        ckpt = torch.load(ckpt_path, map_location="cpu")
        if "module" in ckpt.keys():
            ckpt = ckpt["module"]["model"]
        self.net.load_state_dict(parse_checkpoint(ckpt))

    def prepare_input_scaled(self, image: np.ndarray, current_scale: float, min_scale: float):
        """prepare_input for one entry of a multi-scale test (get_multi_scale_size, base/transforms/utils.py:60-86)."""
        size, center, scale = get_multi_scale_size(image, self.input_size, current_scale, min_scale)
        m = dst_to_src_matrix(center, scale, size)
        raw = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(self.device)
        x = torch.empty((1, 3, size[1], size[0]), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hh_preprocess_u8(raw.data_ptr(), image.shape[0], image.shape[1],
                                                  m.ctypes.data_as(C.POINTER(C.c_double)), x.data_ptr(), size[1], size[0],
                                                  IMAGENET_MEAN.ctypes.data_as(C.POINTER(C.c_float)),
                                                  IMAGENET_STD.ctypes.data_as(C.POINTER(C.c_float)), stream))
        self._keep_raw = raw
        return x, center, scale

    @torch.no_grad()
    def multi_scale_maps(self, raw_image: np.ndarray, scales=(0.5, 1.0, 2.0)):
        """Multi-scale (+ flip if use_flip) test of BASELINE.json configs[3] -- an EXTENSION, the reference has no
        aggregation code.  Every scale runs the same forward/flip-merge as a single-scale call; its two stage heatmaps
        are bilinearly resized to the scale-1 pass's 1/4 and 1/2 resolutions and averaged over the scales
        (hh_resize_accumulate); tags are taken from the scale-1 pass only (as the HigherHRNet paper's test code does
        for the grouping keys).  Returns ([hm_1/4, hm_1/2], tags_list, x_base, center, scale) ready for from_preds."""
        assert 1.0 in scales, "the scale-1 pass provides the tags and the output geometry"
        mn = min(scales)
        xb, center, scale = self.prepare_input_scaled(raw_image, 1.0, mn)
        hms_b, tags_b = self.forward_tta(xb)
        K = self.net.num_kpts
        acc = [torch.empty_like(hms_b[0].contiguous()), torch.empty_like(hms_b[1].contiguous())]
        stream = torch.cuda.current_stream(xb.device).cuda_stream
        wgt = 1.0 / len(scales)
        first = True
        for s in scales:
            hms = hms_b if s == 1.0 else self.forward_tta(self.prepare_input_scaled(raw_image, s, mn)[0])[0]
            for st in range(2):
                src = hms[st]
                _lib.check(self._lib.hh_resize_accumulate(src.data_ptr(), src.stride(0), 1, K, src.shape[2], src.shape[3],
                                                          acc[st].data_ptr(), acc[st].stride(0), acc[st].shape[2], acc[st].shape[3],
                                                          wgt, int(first), stream))
            first = False
        return acc, [t.contiguous() for t in tags_b], xb, center, scale

    def call_multi_scale(self, raw_image: np.ndarray, annot: list | None = None, scales=(0.5, 1.0, 2.0)) -> InferenceKeypointsResult:
        hms, tags, xb, center, scale = self.multi_scale_maps(raw_image, scales)
        self.model_input_shape = tuple(xb.shape[-2:])
        return InferenceKeypointsResult.from_preds(raw_image, annot, xb[0], hms, tags, self.limbs, scale, center, self.det_thr,
                                                   self.tag_thr, self.max_num_people, parser=self._parser)

    def prepare_input(self, image: np.ndarray):
        """model.py:70-76: resize-align -> ToTensor -> Normalize -> [1,3,h,w] on device.  Only the raw uint8 image
        crosses PCIe; warp + normalisation run in hh_preprocess_u8."""
        size, center, scale = get_multi_scale_size(image, self.input_size, 1, 1)
        m = dst_to_src_matrix(center, scale, size)  # destination -> source, as cv2.warpAffine inverts the forward matrix
        raw = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(self.device)
        x = torch.empty((1, 3, size[1], size[0]), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hh_preprocess_u8(raw.data_ptr(), image.shape[0], image.shape[1],
                                                  m.ctypes.data_as(C.POINTER(C.c_double)), x.data_ptr(), size[1], size[0],
                                                  IMAGENET_MEAN.ctypes.data_as(C.POINTER(C.c_float)),
                                                  IMAGENET_STD.ctypes.data_as(C.POINTER(C.c_float)), stream))
        self._keep_raw = raw
        return x, center, scale

    @torch.no_grad()
    def forward_tta(self, x: Tensor) -> tuple[list[Tensor], list[Tensor]]:
        """model.py:79-94 on a batch [B,3,h,w]: net forward (+ flipped pass, un-flip, joint permutation,
        heatmap average). Returns ([hm_1/4, hm_1/2], [tags] or [tags, tags_flipped])."""
        K = self.net.num_kpts
        if not self.use_flip:
            init, dec = self.net.forward_raw(x)
            return [init[:, :K], dec], [init[:, K:]]
        B, _, H, W = x.shape
        stream = torch.cuda.current_stream(x.device).cuda_stream
        # the images and their mirror images as ONE batch of 2B: images of a batch are independent (same bits as two passes), and
        # a forward costs ~2 ms of launch structure whatever the batch -- a single image is a chain of ~350 dependent launches
        x2 = torch.empty((2 * B, 3, H, W), device=x.device, dtype=x.dtype)
        x2[:B].copy_(x)
        _lib.check(self._lib.hh_flip_images(x.data_ptr(), x2[B:].data_ptr(), B, 3, H, W, stream))
        init2, dec2 = self.net.forward_raw(x2)
        init, init_f, dec, dec_f = init2[:B], init2[B:], dec2[:B], dec2[B:]
        tags2 = torch.empty((B, K, H // 4, W // 4), device=x.device, dtype=torch.float32)
        hq, wq = H // 4, W // 4
        p = self._perm.ctypes.data
        # stage 0: heatmaps = init[:, :K] (batch stride 2K planes), tags = init[:, K:]
        _lib.check(self._lib.hh_flip_merge(init.data_ptr(), init.stride(0), init_f.data_ptr(), init_f.stride(0),
                                           init_f[:, K:].data_ptr(), init_f.stride(0), tags2.data_ptr(), tags2.stride(0), p, B, K,
                                           hq, wq, stream))
        _lib.check(self._lib.hh_flip_merge(dec.data_ptr(), dec.stride(0), dec_f.data_ptr(), dec_f.stride(0), None, 0, None, 0, p,
                                           B, K, 2 * hq, 2 * wq, stream))
        return [init[:, :K], dec], [init[:, K:], tags2]

    @torch.no_grad()
    def infer_batch_device(self, x: Tensor):
        """Batched forward + decode entirely on the device -> (joints [B,M,K,3+E], scores [B,M], num_people [B])."""
        hms, tags = self.forward_tta(x)
        return self._parser.decode_batch_device(hms[0], hms[1], tags, adjust=True, refine=True)

    def _on_fast_stream(self, fn):
        """Run `fn` on this model's highest-priority stream, ordered after / before the caller's current stream.  A single
        image is a chain of ~350 small dependent launches; on this stack they follow each other faster on a high-priority
        queue (the engine's branch lanes take the priority of the stream they are forked from)."""
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device, priority=torch.cuda.Stream.priority_range()[1])
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = fn()
        cur.wait_stream(self._stream)
        return out

    @staticmethod
    def _dst_to_src(size, center, scale) -> np.ndarray:
        """The destination -> source 2x3 matrix of the resize-align warp (what cv2.warpAffine inverts for itself)."""
        return dst_to_src_matrix(center, scale, size)

    def _geometry(self, image: np.ndarray):
        """Resize-align geometry of one raw image: ((w, h) of the model input, center, scale, destination->source 2x3 matrix)."""
        size, center, scale = get_multi_scale_size(image, self.input_size, 1, 1)
        return size, center, scale, self._dst_to_src(size, center, scale)

    @torch.no_grad()
    def infer_images(self, raw_images: list[np.ndarray], annots: list | None = None, max_batch: int = 32) -> list[InferenceKeypointsResult]:
        """The batched path behind the reference's single-image interface (`__call__` per image, bin/eval.py:18-49): images are
        bucketed by model-input shape, every bucket runs as batches of up to `max_batch` -- ONE host->device copy of the raw
        uint8 pixels, hh_preprocess_u8 per image into one [B,3,h,w] tensor, one (flip-TTA) forward, one hh_decode, one
        device->host copy of the small result arrays -- and every image gets the result `self(image, annot)` returns
        (same kernels on the same per-image data: images of a batch are independent)."""
        n = len(raw_images)
        annots = annots if annots is not None else [None] * n
        # (size, center, scale) now -- the bucket key --, the warp matrix when the image's batch is staged: the first batch should
        # not wait for n matrix inversions
        geo: list = [get_multi_scale_size(img, self.input_size, 1, 1) for img in raw_images]
        buckets: dict[tuple, list[int]] = {}
        for i, g in enumerate(geo):
            buckets.setdefault(tuple(g[0]), []).append(i)
        results: list = [None] * n
        mean, std = IMAGENET_MEAN.ctypes.data_as(C.POINTER(C.c_float)), IMAGENET_STD.ctypes.data_as(C.POINTER(C.c_float))

        def finish(job):
            """device -> host results of one enqueued batch, un-warp, result objects"""
            chunk, (w, h), x, hms, tags, host_out, done = job
            done.synchronize()
            lists = self._parser.to_lists(*host_out)  # (copies what it returns: the pinned buffers are reused two batches later)
            self.model_input_shape = (h, w)
            for j, i in enumerate(chunk):
                joints, scores = lists[j]
                coords = transform_coords(joints[..., :2], geo[i][1], geo[i][2], (w, h))
                results[i] = InferenceKeypointsResult(raw_images[i], annots[i], x[j], coords, joints[..., 2], joints[..., 3:], scores,
                                                      self.det_thr, self.tag_thr, self.limbs, [t[j:j + 1] for t in hms],
                                                      [t[j:j + 1] for t in tags])

        # Two-deep software pipeline: while the GPU works on batch k the host stages the pixels of batch k+1 into a pinned
        # buffer and only then collects the results of batch k (two staging buffers, each guarded by the event of its last use).
        stage = [None, None]
        stage_free: list = [None, None]
        pending = None
        turn = 0
        for (w, h), idxs in buckets.items():
            for lo in range(0, len(idxs), max_batch):
                chunk = idxs[lo:lo + max_batch]
                sizes = [raw_images[i].size for i in chunk]
                offs = np.cumsum([0] + sizes)
                desc_off = (int(offs[-1]) + 63) // 64 * 64  # the image descriptors travel behind the pixels, in the same copy
                total = desc_off + 64 * len(chunk)
                if stage[turn] is None or stage[turn].numel() < total:
                    stage[turn] = torch.empty(total, dtype=torch.uint8).pin_memory()
                elif stage_free[turn] is not None:
                    stage_free[turn].synchronize()  # the copy that last read this buffer has finished
                host = stage[turn]
                hview = host.numpy()
                descs = hview[desc_off:total].view(_IMAGE_DESC)
                for j, i in enumerate(chunk):
                    img = raw_images[i]
                    np.copyto(hview[offs[j]:offs[j + 1]].reshape(img.shape), img, casting="same_kind")
                    descs[j] = (int(offs[j]), img.shape[0], img.shape[1], self._dst_to_src(*geo[i]).reshape(6))

                # host -> device on a copy stream of its own, so that the pixels of this batch cross PCIe while the previous
                # batch still computes (25 MB per batch of 32 512x512 images: ~1.7 ms that would otherwise sit on the compute stream)
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(self.device)
                    # (the priority of the compute stream: at a lower one the result copies would wait for the whole next batch)
                    self._d2h_stream = torch.cuda.Stream(self.device, priority=torch.cuda.Stream.priority_range()[1])
                with torch.cuda.stream(self._copy_stream):
                    raw = host[:total].to(self.device, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record()

                def run(chunk=chunk, raw=raw, copied=copied, desc_off=desc_off, w=w, h=h, turn=turn):
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(copied)
                    raw.record_stream(cur)
                    x = torch.empty((len(chunk), 3, h, w), device=self.device, dtype=torch.float32)
                    with torch.cuda.device(x.device):  # one launch for the whole batch, whatever the raw sizes
                        _lib.check(self._lib.hh_preprocess_u8_batch(raw.data_ptr(), raw.data_ptr() + desc_off, len(chunk), x.data_ptr(),
                                                                    h, w, mean, std, cur.cuda_stream))
                    hms, tags = self.forward_tta(x)
                    out = self._parser.decode_batch_device(hms[0], hms[1], tags, adjust=True, refine=True)
                    # device -> host into pinned buffers kept per pipeline slot (allocating pinned memory costs ~0.4 ms per array);
                    # finish() of this slot's previous batch ran before this point and copied what it keeps
                    key = (turn, tuple((tuple(t.shape), t.dtype) for t in out))
                    ring = self._out_ring
                    if key not in ring:
                        ring[key] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in out]
                    # ... on a stream of their own behind the decode, so that the next batch's launches do not queue behind them
                    decoded = torch.cuda.Event()
                    decoded.record()
                    with torch.cuda.stream(self._d2h_stream):
                        self._d2h_stream.wait_event(decoded)
                        host_out = [h.copy_(t, non_blocking=True) for h, t in zip(ring[key], out)]
                        done = torch.cuda.Event()
                        done.record()
                    for t in out:
                        t.record_stream(self._d2h_stream)
                    return (chunk, (w, h), x, hms, tags, host_out, done), copied, raw

                # (the model's high-priority stream, for batches as for single images: same-box alternation with the caller's
                # stream, tools/api_throughput.py: 3820 / 3380 against 2600 / 2970 img/s)
                job, copied, _raw = self._on_fast_stream(run)
                stage_free[turn] = copied
                turn ^= 1
                if pending is not None:
                    finish(pending)
                pending = job
        if pending is not None:
            finish(pending)
        return results

    def __call__(self, raw_image: np.ndarray, annot: list | None) -> InferenceKeypointsResult:
        """model.py:78-111"""
        def run():
            x, center, scale = self.prepare_input(raw_image)
            self.model_input_shape = tuple(x.shape[-2:])
            hms, tags = self.forward_tta(x)
            return InferenceKeypointsResult.from_preds(raw_image, annot, x[0], hms, tags, self.limbs, scale, center, self.det_thr,
                                                       self.tag_thr, self.max_num_people, parser=self._parser)
        return self._on_fast_stream(run)
