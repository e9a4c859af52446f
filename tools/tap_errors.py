"""Prints per-tap error of the bf16 engine against the golden fp32 reference activations (GPU box)."""
import importlib, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
g = np.load(os.path.join(REPO, "tests/golden/net_forward.npz"))
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net.to("cuda:0").eval(); net.set_taps(True)
hms, tags = net(torch.from_numpy(pkg.synth.synth_images(1, 64, 64, 0)).cuda())
torch.cuda.synchronize()
taps = net.read_taps()
for k in g.files:
    if k.startswith("w32_64/tap/") and not k.endswith("deconv#1"):
        n = k.split("/tap/")[1]; a, r = taps[n].astype(np.float64), g[k].astype(np.float64)
        print(f"{n:28s} max {np.abs(a-r).max()/np.abs(r).max():.4f} rms {np.sqrt(((a-r)**2).mean())/np.sqrt((r**2).mean()):.4f}")
for n, t in (("hm_q", hms[0]), ("hm_h", hms[1]), ("tags", tags)):
    a, r = t.cpu().numpy().astype(np.float64), g["w32_64/" + n].astype(np.float64)
    print(f"{n:28s} max {np.abs(a-r).max()/np.abs(r).max():.4f} rms {np.sqrt(((a-r)**2).mean())/np.sqrt((r**2).mean()):.4f}")
