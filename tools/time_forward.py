"""Forward-only timing: eager vs hipGraph replay, single-lane vs multi-lane (GPU box).
usage: time_forward.py [B] ["((graph, lanes), ...)"]"""
import ast, importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seq = ast.literal_eval(sys.argv[2]) if len(sys.argv) > 2 else ((False, 0), (True, 0), (False, 1))
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net.cuda().eval()
x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
side = torch.cuda.Stream()
outs = (torch.empty(B, 34, 128, 128, device="cuda"), torch.empty(B, 17, 256, 256, device="cuda"))
lib = pkg._lib.load()
with torch.cuda.stream(side):
    for mode, lanes in seq:
        net.use_graph = mode
        lib.hh_set_multi_lane(net._h, lanes)
        for _ in range(3): net.forward_raw(x, outs)
        side.synchronize()
        n = 20 if B >= 8 else 100
        t0 = time.perf_counter()
        for _ in range(n): net.forward_raw(x, outs)
        side.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B} graph={mode} lanes={lanes}: {dt*1e3:.3f} ms/forward  {B/dt:.0f} img/s  {net.forward_flops(B,512,512)/dt/1e12:.1f} TFLOP/s")
