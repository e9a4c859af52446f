"""Throughput of two full-batch forwards in flight (two engine instances on two highest-priority streams) vs one."""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = 32
def make():
    net = pkg.HigherHRNet(17, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    return net.cuda().eval()
nets = [make(), make()]
x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
outs = [(torch.empty(B, 34, 128, 128, device="cuda"), torch.empty(B, 17, 256, 256, device="cuda")) for _ in range(2)]
prio = torch.cuda.Stream.priority_range()[1]
streams = [torch.cuda.Stream(priority=prio) for _ in range(2)]
torch.cuda.synchronize()
def run(n_in_flight, steps=40):
    for i in range(6):
        k = i % n_in_flight
        with torch.cuda.stream(streams[k]): nets[k].forward_raw(x, outs[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        k = i % n_in_flight
        with torch.cuda.stream(streams[k]): nets[k].forward_raw(x, outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for rep in range(2):
    for n in (1, 2):
        dt = run(n)
        print(f"{n} forward(s) in flight: {dt*1e3:.3f} ms per forward  {B/dt:.0f} img/s", flush=True)
