"""Training-step timing of BASELINE.json configs[2]'s per-GPU share (GPU box): HigherHRNet-W32, batch B @ 512x512,
forward (train-mode BN) + AE loss + backward + Adam, bf16 activations.  python tools/train_bench.py [B] [steps]"""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K, S = 17, 512
net = pkg.HigherHRNet(K, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net = net.cuda().train()
loss_fn = pkg.AEKeypointsLoss()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, **({"fused": True} if os.environ.get("HH_FUSED_ADAM") else {}))
x = torch.from_numpy(pkg.synth.synth_images(B, S, S, 0)).cuda()
hms, masks, joints = pkg.synth.synth_train_targets(B, K, S, 10, seed=0)
hms = [torch.from_numpy(h).cuda() for h in hms]; masks = [torch.from_numpy(m).cuda() for m in masks]
def step():
    ph, pt = net(x)
    hl, push, pull = loss_fn.calculate_loss(ph, pt, hms, masks, joints)
    loss = hl[0] + hl[1] + push[0] + pull[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss
for _ in range(2): l = step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
print(f"train step B={B} @ {S}x{S}: {dt*1e3:.1f} ms/step  {B/dt:.1f} img/s  loss {l.item():.5f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
