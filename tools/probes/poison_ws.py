"""Forward on a NaN-filled workspace (HH_POISON_WS=1) against the same forward on a fresh one: any difference is a read of
workspace bytes that no kernel wrote.  python tools/probes/poison_ws.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
dev = "cuda:0"
def make():
    net = pkg.HigherHRNet(17, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    return net.to(dev).eval()
shapes = [(1, 512, 512), (4, 512, 512), (2, 128, 128), (1, 96, 160), (3, 64, 64)]
xs = [torch.from_numpy(pkg.synth.synth_images(b, h, w, 7 + i)).to(dev) for i, (b, h, w) in enumerate(shapes)]
clean = make()
with torch.no_grad():
    ref = [[t.clone() for t in (lambda o: (o[0][0], o[0][1], o[1]))(clean(x))] for x in xs]
os.environ["HH_POISON_WS"] = "1"
for order in (range(len(xs)), reversed(range(len(xs)))):  # growing and shrinking shapes on one handle
    net = make()
    with torch.no_grad():
        for i in order:
            h, t = net(xs[i])
            got = (h[0], h[1], t)
            nan = [int(torch.isnan(g).sum().item()) for g in got]
            same = [bool(torch.equal(g, r)) for g, r in zip(got, ref[i])]
            print(shapes[i], "NaNs", nan, "bit-identical to the fresh-workspace run", same, flush=True)
