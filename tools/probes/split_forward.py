"""Would two half-batches, software-pipelined against each other, beat one forward of the whole batch?  (GPU box)
Two handles with the same weights; halves of one [32,3,512,512] batch on two streams (each handle forks its own branch lanes), results into
the two halves of the same output tensors; against the plain forward of the whole batch on one handle.  python tools/probes/split_forward.py [B=32] [offset_us]
"""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def make():
    net = pkg.HigherHRNet(17, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    return net.cuda().eval()


whole, a, b = make(), make(), make()
x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
o1 = torch.empty(B, 34, 128, 128, device="cuda")
o2 = torch.empty(B, 17, 256, 256, device="cuda")
r1, r2 = torch.empty_like(o1), torch.empty_like(o2)
h = B // 2
main, sa, sb = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def run_whole():
    with torch.cuda.stream(main):
        whole.forward_raw(x, (r1, r2))


def run_split():
    ev = torch.cuda.Event()
    ev.record(main)
    sa.wait_event(ev)
    sb.wait_event(ev)
    with torch.cuda.stream(sa):
        a.forward_raw(x[:h], (o1[:h], o2[:h]))
    with torch.cuda.stream(sb):
        b.forward_raw(x[h:], (o1[h:], o2[h:]))
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa)
    eb.record(sb)
    main.wait_event(ea)
    main.wait_event(eb)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


run_whole(); run_split(); torch.cuda.synchronize()
print("same bits:", torch.equal(o1, r1), torch.equal(o2, r2))
for _ in range(3):
    print(f"whole batch {timeit(run_whole):.3f} ms   two half-batches side by side {timeit(run_split):.3f} ms", flush=True)
