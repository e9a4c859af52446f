"""Phase stamps of the matcher wave of image 0 (GPU box; needs a library built with -DHH_MATCH_STAMP, tools/probes/match_stamps.sh)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
K, H, W, B = 17, 512, 512, 32
P = int(os.environ.get("HH_DECODE_PEOPLE") or 10)
uniq = [pkg.synth.synth_decode_maps(K, H // 4, W // 4, P, seed=(1000 if P == 10 else 5000) + i)[:3] for i in range(8)]
hm_q = torch.from_numpy(np.stack([uniq[i % 8][0] for i in range(B)])).cuda()
hm_h = torch.from_numpy(np.stack([uniq[i % 8][1] for i in range(B)])).cuda()
tags = torch.from_numpy(np.stack([uniq[i % 8][2][0] for i in range(B)])).cuda()
parser = pkg.MPPEHeatmapParser(K, 30, 0.05, 0.5)
for _ in range(5):
    parser.decode_batch_device(hm_q, hm_h, [tags])
torch.cuda.synchronize()
st = (C.c_ulonglong * (32 * 8))()
assert lib.hh_debug_match_stamps(st) == 0
s = np.array(st, dtype=np.int64).reshape(32, 8)[:K]
names = ["gather candidates", "group means", "cost matrix", "munkres", "(matched mask)", "apply / new groups"]
tot = np.zeros(6)
for it in range(K):
    r = s[it]
    d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5]] if it else [r[1] - r[0], 0, 0, 0, r[5] - r[1], r[6] - r[5]]
    tot += np.array(d)
    print(f"joint {it:2d}: " + "  ".join(f"{n} {int(x)}" for n, x in zip(names, d)) + f"   | total {int(r[6] - r[0])}")
print("sum over joints (cycles):", {n: int(x) for n, x in zip(names, tot)}, " wave total", int(s[K - 1][6] - s[0][0]))
