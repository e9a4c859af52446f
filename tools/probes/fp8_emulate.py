"""CPU emulation of e4m3 quantisation policies on the oracle forward (test infrastructure, runs anywhere):
    python tools/probes/fp8_emulate.py [C=32] [size=128]
Where does the fp8 path's output error come from, and what does a bf16 residual trunk buy?  The oracle's fp32 walk is re-run
with fake quantisation (torch.float8_e4m3fn casts, per-tensor scales amax/240, per-cout weight scales max->448):
  A  every conv output and every block output e4m3 (the round-2 engine: the residual trunk is re-quantised by every block)
  B  conv INPUTS e4m3, block outputs / residual trunk / fusion sums kept in bf16 (round 3: two representations per trunk tensor)
  B+ B with stem, stage 0 and the three 1x1 / transposed heads in bf16 (fp8 only in the BasicBlocks and fusion convs)
  W  weights only (activations fp32)
"""
import importlib, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import forward as ofw
pkg = importlib.import_module("pytorch-human-pose_amd")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.set_num_threads(8)

def q8(t, target=240.0):
    s = t.abs().max().clamp_min(1e-30) / target
    return (t / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * s
def qw(w, transposed=False):
    dims = (0, 2, 3) if transposed else (1, 2, 3)
    s = w.abs().amax(dim=dims, keepdim=True).clamp_min(1e-30) / 448.0
    return (w / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * s
def b16(t): return t.to(torch.bfloat16).to(torch.float32)

MODE = "A"
def fold(p, conv, bn):
    w = p[f"{conv}.weight"]; b = p.sub(bn)
    sc = b["weight"] / torch.sqrt(b["running_var"] + 1e-5)
    return w * sc[:, None, None, None], b["bias"] - b["running_mean"] * sc

def conv_bn(x, p, conv, bn, stride=1, relu=False, lowp=True, out_trunk=False):
    """x: the tensor as the previous op left it.  lowp: this conv runs on the fp8 MFMA path."""
    w, sh = fold(p, conv, bn)
    pad = (w.shape[-1] - 1) // 2
    if lowp and MODE != "W": x = q8(x)
    if lowp: w = qw(w)
    else: x, w = b16(x), b16(w)
    y = F.conv2d(x, w, sh, stride, pad)
    return F.relu(y) if relu else y

def store(y, trunk):
    """what the producing op writes: A -> e4m3 always; B -> bf16 (consumers that are convs quantise on read)"""
    if MODE == "A": return q8(y)
    if MODE == "W": return y
    return b16(y)

STAGE0_E4M3 = False  # stage 0's 256-channel trunk e4m3 only (no bf16 twin): 3x less HBM traffic in its HBM-bound 1x1 convs
def bottleneck(x, p, lowp):
    st = (lambda t, trunk: q8(t)) if (STAGE0_E4M3 and MODE == "B") else store
    y = st(conv_bn(x, p, "conv1", "bn1", relu=True, lowp=lowp), False)
    y = st(conv_bn(y, p, "conv2", "bn2", relu=True, lowp=lowp), False)
    y = conv_bn(y, p, "conv3", "bn3", lowp=lowp)
    r = st(conv_bn(x, p, "downsample.0", "downsample.1", lowp=lowp), False) if p.has("downsample.0.weight") else x
    return st(F.relu(y + r), True)

def basic(x, p, lowp=True):
    y = store(conv_bn(x, p, "conv1", "bn1", relu=True, lowp=lowp), False)
    y = conv_bn(y, p, "conv2", "bn2", lowp=lowp)
    return store(F.relu(y + x), True)

def fusion(xs, p, n_out):
    outs = []
    for i in range(n_out):
        acc = 0
        for j, x in enumerate(xs):
            q = p.sub(f"scales_fusion_layers.{i}.{j}")
            if j == i: t = x
            elif j > i:
                t = store(conv_bn(x, q, "0", "1"), False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = x
                for k in range(i - j):
                    t = conv_bn(t, q.sub(str(k)), "0", "1", stride=2, relu=(k != i - j - 1))
                    if k != i - j - 1: t = store(t, False)
                    elif MODE == "A": t = q8(t)  # (the engine accumulates through the e4m3 output tensor)
            acc = acc + t
        outs.append(store(F.relu(acc), True))
    return outs

def forward(images, sd, K=17, hi_prec=False, hi_heads=None):
    p = ofw._SD(sd).sub("backbone")
    lp0 = not hi_prec
    hi_heads = hi_prec if hi_heads is None else hi_heads
    x = F.relu(F.conv2d(b16(images), b16(fold(p, "conv1", "bn1")[0]), fold(p, "conv1", "bn1")[1], 2, 1))  # stem conv1: bf16 operands in every mode
    x = store(x, False)
    x = store(conv_bn(x, p, "conv2", "bn2", stride=2, relu=True, lowp=lp0), False)
    xs = [x]
    nblocks = [1, 1, 4, 3]
    for s in range(4):
        sp = p.sub(f"stages.{s}")
        for b in range(nblocks[s]):
            bp = sp.sub(f"blocks.{2 * b}")
            new = []
            for i, t in enumerate(xs):
                for u in range(4):
                    t = bottleneck(t, bp.sub(f"scales_blocks.{i}.{u}"), lp0) if s == 0 else basic(t, bp.sub(f"scales_blocks.{i}.{u}"))
                new.append(t)
            xs = new
            last = s == 3 and b == nblocks[s] - 1
            xs = fusion(xs, sp.sub(f"blocks.{2 * b + 1}"), 1 if last else len(xs)) if s > 0 else xs
        if s < 3:
            tp = sp.sub("transition_layer.transition_blocks")
            n = len(xs)
            new = store(conv_bn(xs[-1], tp.sub(str(n)), "0", "1", stride=2, relu=True, lowp=(lp0 or s > 0)), True)
            if s == 0: xs = [store(conv_bn(xs[0], tp.sub("0"), "0", "1", relu=True, lowp=lp0), True)]
            xs = xs + [new]
    feats = xs[0]
    pp = ofw._SD(sd)
    def head(x, w, b, transposed=False, bn=None):
        if not hi_heads:
            if MODE != "W": x = q8(x)
            w = qw(w, transposed)
        else: x, w = b16(x), b16(w)
        return F.conv_transpose2d(x, w, None, 2, 1, 0) if transposed else F.conv2d(x, w, b)
    init = head(feats, pp["init_heatmaps_head.weight"], pp["init_heatmaps_head.bias"])
    d = pp.sub("deconv_layers.0")
    y = torch.cat((feats, store(init, False)), 1)
    bn = d.sub("deconv.1"); sc = bn["weight"] / torch.sqrt(bn["running_var"] + 1e-5)
    y = head(y, d["deconv.0.weight"] * sc[None, :, None, None], None, True) + (bn["bias"] - bn["running_mean"] * sc)[None, :, None, None]
    y = store(F.relu(y), True)
    for r in range(4): y = basic(y, d.sub(f"resid_blocks.{r}"))
    out = head(y, d["final_layer.weight"], d["final_layer.bias"])
    return [init[:, :K], out[:, :K]], init[:, K:]

net = pkg.HigherHRNet(17, C)
sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 1)) for k, v in net.state_dict().items()}
x = torch.from_numpy(pkg.synth.synth_images(1, S, S, 1))
with torch.no_grad():
    rh, rt = ofw.higher_hrnet(x, sd, 17)
    def rep(tag, hms, tags):
        e = lambda a, b: (float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt()), float((a - b).abs().max() / b.abs().max()))
        print(f"{tag:4s} hm_q rms {e(hms[0], rh[0])[0]:.4f} max {e(hms[0], rh[0])[1]:.4f} | hm_h rms {e(hms[1], rh[1])[0]:.4f} max {e(hms[1], rh[1])[1]:.4f} | tags rms {e(tags, rt)[0]:.4f} max {e(tags, rt)[1]:.4f}", flush=True)
    for mode, hp, hh, s0 in (("W", False, False, False), ("A", False, False, False), ("B", False, False, False), ("B", True, True, False), ("B", False, True, False), ("B", True, False, False), ("B", False, True, True)):
        MODE = mode
        STAGE0_E4M3 = s0
        rep(mode + ("s" if hp else "") + ("h" if hh else "") + ("0" if s0 else ""), *forward(x, sd, 17, hp, hh))
