"""Train-mode BatchNorm passes alone: effective HBM rate of hh_bn_train_forward / hh_bn_train_backward at the layer shapes of the
W32 training step (batch 32 @ 512x512).  python tools/probes/bn_probe.py   (HH_LIB=... for another build of the library)"""
import importlib, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
ops = importlib.import_module("pytorch-human-pose_amd.keypoints.train_ops")

dev = torch.device("cuda:0")
shapes = [(32, 32, 128, 128), (32, 64, 128, 128), (32, 256, 128, 128), (32, 64, 64, 64), (32, 128, 32, 32), (32, 256, 16, 16), (32, 64, 256, 256), (32, 32, 256, 256)]
reps = 30
for B, C, H, W in shapes:
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = torch.randn((B, C, H, W), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn((B, C, H, W), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, C, H, W), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.1
    nbytes = x.numel() * 2
    out = []
    for with_res in (False, True):
        r = res if with_res else None
        for _ in range(3):
            y, mean, invstd = ops.bn_train_forward(x, gamma, beta, 1e-5, r, True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(reps):
            y, mean, invstd = ops.bn_train_forward(x, gamma, beta, 1e-5, r, True)
        e1.record()
        for _ in range(reps):
            if with_res or os.environ.get("BN_PROBE_KEEP_Y"):
                ops.bn_train_backward(x, y, dy, mean, invstd, gamma, True, want_dres=with_res)
            else:  # the form the training step uses without a residual: y is not read
                ops.bn_train_backward(x, None, dy, mean, invstd, gamma, True, beta=beta)
        e2.record()
        torch.cuda.synchronize()
        tf, tb = e0.elapsed_time(e1) / reps * 1e3, e1.elapsed_time(e2) / reps * 1e3
        fb = nbytes * (3 + with_res)            # stats read + normalise read (+ residual) + write
        bb = nbytes * ((7 + with_res) if with_res or os.environ.get("BN_PROBE_KEEP_Y") else 5)  # (x, y, dy) twice + dx (+ dres); plain: (x, dy) twice + dx
        out.append(f"{'res' if with_res else 'plain'}: fwd {tf:7.1f} us {fb / tf / 1e6:6.2f} TB/s | bwd {tb:7.1f} us {bb / tb / 1e6:6.2f} TB/s")
    print(f"B{B} C{C:<3d} {H}x{W} ({nbytes / 1e6:6.1f} MB)  " + "  ||  ".join(out), flush=True)
