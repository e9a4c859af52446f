"""Experiment: one B=32 forward vs two concurrent B=16 forwards (two engines, two caller streams)."""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
def make():
    net = pkg.HigherHRNet(17, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    return net.cuda().eval()
n1, n2 = make(), make()
x = torch.from_numpy(pkg.synth.synth_images(32, 512, 512, 0)).cuda()
xa, xb = x[:16].contiguous(), x[16:].contiguous()
o32 = (torch.empty(32, 34, 128, 128, device="cuda"), torch.empty(32, 17, 256, 256, device="cuda"))
oa = (torch.empty(16, 34, 128, 128, device="cuda"), torch.empty(16, 17, 256, 256, device="cuda"))
ob = (torch.empty(16, 34, 128, 128, device="cuda"), torch.empty(16, 17, 256, 256, device="cuda"))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run32():
    with torch.cuda.stream(s1): n1.forward_raw(x, o32)
def runseq16():  # two half batches one after the other on one stream: do the smaller tensors stay in the 256 MB Infinity Cache?
    with torch.cuda.stream(s1):
        n1.forward_raw(xa, oa)
        n1.forward_raw(xb, ob)
x8 = [x[i * 8:(i + 1) * 8].contiguous() for i in range(4)]
o8 = [(torch.empty(8, 34, 128, 128, device="cuda"), torch.empty(8, 17, 256, 256, device="cuda")) for _ in range(4)]
def runseq8():
    with torch.cuda.stream(s1):
        for i in range(4): n1.forward_raw(x8[i], o8[i])
def run2x16():
    with torch.cuda.stream(s1): n1.forward_raw(xa, oa)
    with torch.cuda.stream(s2): n2.forward_raw(xb, ob)
for lanes in (1, 0):
    lib.hh_set_multi_lane(n1._h, lanes); lib.hh_set_multi_lane(n2._h, lanes)
    for name, fn in (("1 x B32", run32), ("2 x B16 concurrent", run2x16), ("2 x B16 sequential", runseq16), ("4 x B8 sequential", runseq8)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        print(f"lanes={lanes} {name}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per 32 images")
