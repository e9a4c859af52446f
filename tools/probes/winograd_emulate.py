"""CPU emulation of Winograd F(2x2, 3x3) with bf16 operands for the wide 3x3 stride-1 convolutions (test
infrastructure, runs anywhere):   python tools/probes/winograd_emulate.py [min_cin=128]

Question: if the 128- / 256-channel 3x3 convolutions (W32; 192 / 384 for W48) ran as Winograd F(2x2, 3x3) on the bf16 MFMA path --
transformed weights U = G g G^T and transformed input tiles V = B^T d B rounded to bf16, products accumulated in fp32, output
transform in fp32 -- would every output still meet the tolerances the direct bf16 path is tested with (tests/test_gpu_parity.py::_close:
max |err| <= 5 % of max |ref|, rms err <= 2 % of rms ref)?  The fp32 oracle walk is re-run twice with the engine's roundings emulated
(BatchNorm folded into the weights before they are rounded to bf16, every conv input rounded to bf16, fp32 accumulation, every stored
activation rounded to bf16): once with all convolutions direct, once with the wide 3x3 ones through the transform.
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import forward as ofw  # noqa: E402

pkg = importlib.import_module("pytorch-human-pose_amd")
MIN_CIN = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.set_num_threads(8)

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def b16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def winograd_conv(x, w, lowp=True):
    """3x3 stride-1 pad-1 convolution as F(2x2, 3x3); x [B,C,H,W] (H, W even), w [O,C,3,3]."""
    B_, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)  # [B,C,H/2,W/2,4,4]
    V = Bt @ d @ Bt.t()
    U = G @ w @ G.t()  # [O,C,4,4]
    if lowp:
        V, U = b16(V), b16(U)
    M = torch.einsum("ocij,bchwij->bohwij", U, V)
    Y = At @ M @ At.t()  # [B,O,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], H, W)


MODE = "direct"
N_WINO = 0


def conv_bn(x, p, conv, bn, stride=1, pad=None, relu=False):
    """ofw._conv_bn with the engine's roundings: BN folded, operands bf16, fp32 accumulate, the stored result bf16."""
    global N_WINO
    w = p[f"{conv}.weight"]
    b = p.sub(bn)
    sc = b["weight"] / torch.sqrt(b["running_var"] + ofw.EPS)
    wf, sh = b16(w * sc[:, None, None, None]) if MODE != "wino_fold_after" else w * sc[:, None, None, None], b["bias"] - b["running_mean"] * sc
    if pad is None:
        pad = (w.shape[-1] - 1) // 2
    xq = b16(x)
    if MODE.startswith("wino") and w.shape[-1] == 3 and stride == 1 and w.shape[1] >= MIN_CIN and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0:
        y = winograd_conv(xq, w * sc[:, None, None, None]) + sh[None, :, None, None]  # (U is made from the fp32 folded weights and rounded once)
        N_WINO += 1
    else:
        y = F.conv2d(xq, wf, sh, stride, pad)
    return F.relu(y) if relu else y


def run(mode, x, sd):
    global MODE, N_WINO
    MODE, N_WINO = mode, 0
    orig = ofw._conv_bn
    ofw._conv_bn = conv_bn
    try:
        with torch.no_grad():
            hms, tags = ofw.higher_hrnet(x, sd, 17)
    finally:
        ofw._conv_bn = orig
    return [t.numpy() for t in (*hms, tags)], N_WINO


def main():
    # sanity: the transform itself is exact in fp32
    xt, wt = torch.randn(1, 8, 8, 8), torch.randn(4, 8, 3, 3)
    assert (winograd_conv(xt, wt, lowp=False) - F.conv2d(xt, wt, None, 1, 1)).abs().max() < 1e-4
    for C, B_, H, W, seed in ((32, 2, 128, 128, 1), (32, 1, 96, 160, 2), (48, 1, 64, 64, 3), (32, 1, 256, 256, 5)):
        net = pkg.HigherHRNet(17, C)
        sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, seed)) for k, v in net.state_dict().items()}
        x = torch.from_numpy(pkg.synth.synth_images(B_, H, W, seed))
        with torch.no_grad():
            hms, tags = ofw.higher_hrnet(x, sd, 17)
        ref = [t.numpy() for t in (*hms, tags)]
        print(f"W{C} {B_}x{H}x{W}:")
        for mode in ("direct", "wino"):
            got, nw = run(mode, x, sd)
            errs = []
            for name, g, r in zip(("hm_q", "hm_h", "tags"), got, ref):
                mx = np.abs(g - r).max() / np.abs(r).max()
                rms = np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r ** 2))
                errs.append(f"{name} max {mx * 100:.2f} % rms {rms * 100:.2f} %")
            print(f"  {mode:7s} ({nw:3d} convs through the transform): " + "   ".join(errs), flush=True)


if __name__ == "__main__":
    main()
