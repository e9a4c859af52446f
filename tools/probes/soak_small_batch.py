"""Repeat the B = 1 / B = 4 forwards of test_forward_full_size_samples_and_batch_consistency and count runs whose bits differ from
the first one (a race in the multi-lane schedule would show here): python tools/probes/soak_small_batch.py [iters]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch-human-pose_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda:0"
for env in ({}, {"HH_FULL_JOIN": "1"}, {"HH_NO_JUNC_PAIR": "1"}):
    os.environ.update(env)
    net = pkg.HigherHRNet(17, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    net = net.to(dev).eval()
    for k in env: del os.environ[k]
    x1 = torch.from_numpy(pkg.synth.synth_images(1, 512, 512, 7)).to(dev)
    xb = torch.from_numpy(pkg.synth.synth_images(4, 512, 512, 8)).to(dev)
    xb[0] = x1[0]; xb[3] = x1[0]
    ref1 = ref4 = None
    bad = {"b1": 0, "b4": 0, "slot": 0}
    worst = 0.0
    with torch.no_grad():
        for it in range(iters):
            h1, t1 = net(x1)
            h4, t4 = net(xb)
            cur1 = [h1[0].clone(), h1[1].clone(), t1.clone()]
            cur4 = [h4[0].clone(), h4[1].clone(), t4.clone()]
            if ref1 is None: ref1, ref4 = cur1, cur4
            if not all(torch.equal(a, b) for a, b in zip(cur1, ref1)):
                bad["b1"] += 1; worst = max(worst, max((a - b).abs().max().item() for a, b in zip(cur1, ref1)))
            if not all(torch.equal(a, b) for a, b in zip(cur4, ref4)):
                bad["b4"] += 1; worst = max(worst, max((a - b).abs().max().item() for a, b in zip(cur4, ref4)))
            if not (torch.equal(h4[0][0], h1[0][0]) and torch.equal(h4[1][3], h1[1][0]) and torch.equal(t4[3], t1[0])): bad["slot"] += 1
    print(env or "default", "runs that differ:", bad, "of", iters, "largest difference", worst, flush=True)
