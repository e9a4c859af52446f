"""Does a highest-priority caller stream shorten the small-batch forward (a chain of ~300 dependent launches)?"""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net.cuda().eval()
lib = pkg._lib.load()
for B in (1, 8):
    x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
    outs = (torch.empty(B, 34, 128, 128, device="cuda"), torch.empty(B, 17, 256, 256, device="cuda"))
    for prio in (0, -1, 0, -1):
        st = torch.cuda.Stream(priority=prio)
        with torch.cuda.stream(st):
            for graph, lanes in ((True, 0), (False, 1)):
                net.use_graph = graph
                lib.hh_set_multi_lane(net._h, lanes)
                for _ in range(5): net.forward_raw(x, outs)
                st.synchronize()
                t0 = time.perf_counter()
                for _ in range(50): net.forward_raw(x, outs)
                st.synchronize()
                print(f"B={B} caller priority {prio:2d} graph={int(graph)} lanes={lanes}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
