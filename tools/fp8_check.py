"""fp8 engine against the fp32 goldens, tap by tap (GPU box): python tools/fp8_check.py [C] [size]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch-human-pose_amd")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "net_forward.npz"))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tag, seed = {32: ("w32_64", 0), 48: ("w48_64", 3)}[C]
dev = "cuda:0"
def err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6), np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-6)
VARIANTS = [("bf16", {}), ("fp8", {}), ("fp8", {"HH_FP8_HEADS": "e4m3"}), ("fp8", {"HH_FP8_TRUNK": "e4m3", "HH_FP8_HEADS": "e4m3"})]
for dtype, env in VARIANTS:
    os.environ.update(env)
    net = pkg.HigherHRNet(17, C, dtype=dtype)
    for k in env: del os.environ[k]
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, seed)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    x = torch.from_numpy(pkg.synth.synth_images(1, 64, 64, seed)).to(dev)
    if dtype == "fp8":
        cal = torch.from_numpy(pkg.synth.synth_images(4, 64, 64, 100)).to(dev)
        net.calibrate(torch.cat([cal, x]))
    net.set_taps(True)
    hms, tags = net(x)
    torch.cuda.synchronize()
    taps = net.read_taps()
    print("==", dtype, env or "")
    for k in g.files:
        if k.startswith(tag + "/tap/"):
            name = k.split("/tap/")[1]
            if name in taps and taps[name].shape == g[k].shape:
                print(f"  {name:28s} max {err(taps[name], g[k])[0]:.4f} rms {err(taps[name], g[k])[1]:.4f}")
    for name, t in (("hm_q", hms[0]), ("hm_h", hms[1]), ("tags", tags)):
        e = err(t.cpu().numpy(), g[f"{tag}/{name}"])
        print(f"  OUT {name:24s} max {e[0]:.4f} rms {e[1]:.4f}")
