"""Forward-only timing: eager launches vs hipGraph replay (GPU box)."""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net.cuda().eval()
x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for mode in (False, True, False, True):
        net.use_graph = mode
        for _ in range(3): out = net.forward_raw(x)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): out = net.forward_raw(x)
        side.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"graph={mode}: {dt*1e3:.3f} ms/forward  {B/dt:.0f} img/s  {net.forward_flops(B,512,512)/dt/1e12:.1f} TFLOP/s")
