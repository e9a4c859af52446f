"""Per-layer live timing of one forward (device-clock probe inside every conv launch): time, TFLOP/s."""
import ctypes as C, importlib, os, sys, collections, re
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = pkg.HigherHRNet(17, 32)
net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
net.cuda().eval()
x = torch.from_numpy(pkg.synth.synth_images(B, 512, 512, 0)).cuda()
lib = pkg._lib.load()
net.use_graph = False
for _ in range(3): net.forward_raw(x)
torch.cuda.synchronize()
lib.hh_profile_enable(net._h, 1)
R = 3  # 3 x ~300 launches stay inside the HH_PROF_SLOTS device-clock slots
for _ in range(R): net.forward_raw(x)
torch.cuda.synchronize()
n = lib.hh_profile_count(net._h)
cfg, fl, ms, name = C.c_int(), C.c_double(), C.c_float(), C.c_char_p()
agg = collections.OrderedDict()
for i in range(n):
    kms, by = C.c_float(), C.c_double()
    lib.hh_profile_get(net._h, i, C.byref(cfg), C.byref(fl), C.byref(by), C.byref(ms), C.byref(kms), C.byref(name))
    nm = name.value.decode()
    key = re.sub(r"blocks\.\d+\.scales_blocks\.(\d+)\.\d+", r"blocks.*.scales_blocks.\1.*", nm)
    key = re.sub(r"blocks\.\d+\.scales_fusion", "blocks.*.scales_fusion", key)
    key = re.sub(r"resid_blocks\.\d+", "resid_blocks.*", key)
    d = agg.setdefault((key, cfg.value), [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += (kms.value if kms.value > 0 else ms.value); d[2] += fl.value; d[3] += by.value
tot = sum(v[1] for v in agg.values()) / R
print(f"conv total {tot:.3f} ms per forward (B={B}), {sum(v[2] for v in agg.values())/R/tot/1e9:.1f} TFLOP/s")
cv = (C.c_int * 7)()
for (key, c), (cnt, t, f, by_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lib.hh_conv_config(c, cv)
    ai = f / by_
    ceil = min(2500.0, ai * 8.0)  # roofline ceiling in TFLOP/s: min(MFMA peak, AI x 8 TB/s)
    print(f"{t/R:8.3f} ms  {cnt//R:3d}x {t/cnt*1e3:8.1f} us  {f/t/1e9:7.1f} TF/s  {by_/t/1e9:6.2f} TB/s  AI {ai:5.0f}  ceiling {ceil:6.0f} ({f/t/1e9/ceil*100:3.0f}%)  cfg{tuple(cv) if c < 100 else c}  {key}")
