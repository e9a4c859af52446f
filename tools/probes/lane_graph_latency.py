"""Forward latency of the multi-lane plan: eager launches vs replay of the explicit hipGraph (build_lane_graph), and bit equality
of the two, at batch 1 / 4 / 8 / 32 (GPU box): python tools/probes/lane_graph_latency.py"""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
DEV = "cuda:0"


def make(env):
    os.environ.update(env)
    net = pkg.HigherHRNet(17, 32)
    for k in env: del os.environ[k]
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    return net.to(DEV).eval()


def timeit(fn, n=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


eager, graph = make({"HH_LANE_GRAPH": "0"}), make({"HH_LANE_GRAPH": "1"})
prio = torch.cuda.Stream.priority_range()[1]
side = torch.cuda.Stream(DEV, priority=prio)
for B in (1, 2, 4, 8, 32):
    x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, B)).to(DEV)
    outs_e = (torch.empty(B, 34, 128, 128, device=DEV), torch.empty(B, 17, 256, 256, device=DEV))
    outs_g = (torch.empty(B, 34, 128, 128, device=DEV), torch.empty(B, 17, 256, 256, device=DEV))
    with torch.cuda.stream(side):
        me = timeit(lambda: eager.forward_raw(x, outs_e))
        mg = timeit(lambda: graph.forward_raw(x, outs_g))
        me2 = timeit(lambda: eager.forward_raw(x, outs_e))
        mg2 = timeit(lambda: graph.forward_raw(x, outs_g))
    torch.cuda.synchronize()
    same = torch.equal(outs_e[0], outs_g[0]) and torch.equal(outs_e[1], outs_g[1])
    print(f"B={B:2d} 512x512 forward: eager lanes {me:.3f} / {me2:.3f} ms, explicit graph {mg:.3f} / {mg2:.3f} ms, bit-equal {same}", flush=True)
