"""Race soak at the headline shape: 200 multi-lane forwards of B=32 @ 512x512 on alternating inputs, every output compared bit for
bit with the single-stream plan's (a missing cross-lane edge shows up as a run-to-run difference)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net = net.cuda().eval()
xs = [torch.from_numpy(pkg.synth.synth_images(32, 512, 512, s)).cuda() for s in (0, 1)]
lib.hh_set_multi_lane(net._h, 0)
refs = [[t.clone() for t in net.forward_raw(x)] for x in xs]
lib.hh_set_multi_lane(net._h, 1)
s = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
bad = 0
with torch.cuda.stream(s):
    for i in range(200):
        out = net.forward_raw(xs[i & 1])
        if not all(torch.equal(a, b) for a, b in zip(out, refs[i & 1])):
            bad += 1
s.synchronize()
print("mismatching forwards:", bad, "of 200")
sys.exit(1 if bad else 0)
