"""Weight gradients of a set of layer shapes to an .npz (bit comparison of two builds of the library):
HH_LIB=<lib> python tools/probes/wgrad_bits.py out.npz ; python tools/probes/wgrad_bits.py --compare a.npz b.npz"""
import importlib, os, sys
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
    print("compared", len(a.files), "gradients:", "all bit-identical" if not bad else f"DIFFERENT: {bad}")
    sys.exit(1 if bad else 0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
ops = importlib.import_module("pytorch-human-pose_amd.keypoints.train_ops")
dev = torch.device("cuda:0")
out = {}
g = torch.Generator().manual_seed(3)
for (B, cin, cout, ks, stride, hw) in [(32, 32, 32, 3, 1, 128), (32, 64, 64, 3, 1, 64), (32, 128, 128, 3, 1, 32), (32, 256, 256, 3, 1, 16), (32, 256, 64, 1, 1, 128),
                                       (32, 64, 256, 1, 1, 128), (32, 32, 64, 3, 2, 128), (8, 48, 48, 3, 1, 40), (3, 64, 128, 3, 2, 24), (2, 136, 72, 1, 1, 16)]:
    x = torch.randn(B, cin, hw, hw, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, hw // stride, hw // stride, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out[f"{B}_{cin}_{cout}_{ks}_{stride}_{hw}"] = ops.conv2d_weight_grad(x, dy, ks, stride).cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "gradients")
