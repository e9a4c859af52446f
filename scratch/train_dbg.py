import sys, importlib, numpy as np, torch
sys.path.insert(0,'.')
pkg=importlib.import_module("pytorch-human-pose_amd")
from oracle import forward as ofw
torch.manual_seed(0)
K,C=17,32
net=pkg.HigherHRNet(K,C)
sd={k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 5)) for k,v in net.state_dict().items()}
net.load_state_dict(sd); net=net.cuda().train()
x=torch.from_numpy(pkg.synth.synth_images(2,128,128,seed=1))
# reference: oracle in train mode, fp32 CPU, autograd
ref_sd={k:(v.clone().float().requires_grad_() if v.dtype.is_floating_point and not k.endswith(('running_mean','running_var')) else v.clone()) for k,v in sd.items()}
hms,tags=ofw.higher_hrnet(x, ref_sd, K, train=True)
loss_ref=(hms[0]**2).mean()+(hms[1]**2).mean()+(tags**2).mean()
loss_ref.backward()
h2,t2=net(x.cuda())
loss=(h2[0]**2).mean()+(h2[1]**2).mean()+(t2**2).mean()
loss.backward()
def rel(a,b): return float((a.float().cpu()-b).abs().max()/b.abs().max().clamp_min(1e-12))
print("loss",float(loss_ref),float(loss), "hm0",rel(h2[0].detach(),hms[0].detach()),"hm1",rel(h2[1].detach(),hms[1].detach()),"tags",rel(t2.detach(),tags.detach()))
worst=[]
for name,p in net.named_parameters():
    g=p.grad
    r=ref_sd[name].grad
    if g is None or r is None: print("nograd",name, g is None, r is None); continue
    g=g.float().cpu().flatten(); r=r.flatten()
    cos=float(torch.dot(g,r)/(g.norm()*r.norm()+1e-30)); ratio=float(g.norm()/(r.norm()+1e-30))
    worst.append((cos,ratio,name,float(r.norm())))
worst.sort()
for w in worst[:12]: print("cos %.4f ratio %.3f %s |ref| %.3e"%w)
print("median cos", np.median([w[0] for w in worst]), "n",len(worst), "cos<0.95:", sum(w[0]<0.95 for w in worst))
