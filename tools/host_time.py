"""Host-side enqueue time of one forward (eager, branch lanes) vs its GPU time (GPU box)."""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net = net.cuda().eval()
for B in (32, 8, 1):
    x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
    outs = (torch.empty(B, 34, 128, 128, device="cuda"), torch.empty(B, 17, 256, 256, device="cuda"))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for lanes in (1, 0):
            lib.hh_set_multi_lane(net._h, lanes); net.use_graph = False
            for _ in range(5): net.forward_raw(x, outs)
            torch.cuda.synchronize()
            host = []
            t0 = time.perf_counter()
            for _ in range(20):
                t = time.perf_counter(); net.forward_raw(x, outs); host.append(time.perf_counter() - t)
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) / 20
            print(f"B={B} lanes={lanes}: host enqueue {np.median(host)*1e3:.3f} ms per forward, wall {tot*1e3:.3f} ms per forward")
