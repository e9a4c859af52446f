"""Determinism soak of the whole step: 100 x (multi-lane forward + decode of its own outputs) on a fixed pass-through batch; joints,
scores and people counts must repeat bit for bit, and equal the single-stream plan's."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
K = 17
net = pkg.HigherHRNet(K, 32)
net.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.synth_passthrough_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, K, 0, tag_gain=8.0).items()})
net = net.cuda().eval()
x = torch.from_numpy(pkg.synth.synth_passthrough_images(32, 128, 128, 10, K, 0, tag_gain=8.0)[0]).cuda()
parser = pkg.MPPEHeatmapParser(K, 30, 0.05, 0.5)
def step():
    init, dec = net.forward_raw(x)
    return [t.clone() for t in parser.decode_batch_device(init[:, :K], dec, [init[:, K:]], adjust=True, refine=True)]
lib.hh_set_multi_lane(net._h, 0)
ref = step()
lib.hh_set_multi_lane(net._h, 1)
s = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
bad = 0
with torch.cuda.stream(s):
    for i in range(100):
        out = step()
        bad += not all(torch.equal(a, b) for a, b in zip(out, ref))
s.synchronize()
print("people decoded:", int(ref[2].sum().item()), " steps that differ from the single-stream step:", bad, "of 100")
sys.exit(1 if bad else 0)
