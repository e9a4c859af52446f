"""Can the whole training step (forward, AE loss, backward, Adam) be captured in a hipGraph and replayed?"""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, S = 17, 512
net = pkg.HigherHRNet(K, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net = net.cuda().train()
loss_fn = pkg.AEKeypointsLoss()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)
x = torch.from_numpy(pkg.synth.synth_images(B, S, S, 0)).cuda()
hms, masks, joints = pkg.synth.synth_train_targets(B, K, S, 10, seed=0)
hms = [torch.from_numpy(h).cuda() for h in hms]; masks = [torch.from_numpy(m).cuda() for m in masks]
joints = [importlib.import_module("pytorch-human-pose_amd.keypoints.loss").upload_joints(list(joints[0]), K, S // 4, S // 4, "cuda")] + list(joints[1:])
def step():
    ph, pt = net(x)
    hl, push, pull = loss_fn.calculate_loss(ph, pt, hms, masks, joints)
    loss = hl[0] + hl[1] + push[0] + pull[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): l = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", l.item(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(2): g.replay()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"graph replay: {dt*1e3:.1f} ms/step  {B/dt:.1f} img/s  loss {static_loss.item():.5f}")
