"""Micro-benchmark of single conv shapes through hh_debug_conv_bench (GPU box).
usage: conv_bench.py [case ...]   case = cfg,B,H,W,cin,cout,res"""
import ctypes as C, importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (0, 32, 128, 128, 32, 32, 1), (0, 32, 256, 256, 32, 32, 1), (1, 32, 64, 64, 64, 64, 1), (1, 32, 32, 32, 128, 128, 1),
    (2, 32, 16, 16, 256, 256, 1), (1, 32, 128, 128, 64, 64, 0), (9, 32, 128, 128, 64, 256, 1), (9, 32, 128, 128, 256, 64, 0)]
cv = (C.c_int * 7)()
for cfg, B, H, W, cin, cout, res in cases:
    ms = C.c_float()
    st = np.zeros(16, np.uint64)
    md = C.c_float(-1)
    ref = 1 if cin % 32 == 0 else 3
    pkg._lib.check(lib.hh_debug_conv_bench(cfg, B, H, W, cin, cout, res, 1, 50, C.byref(ms), st.ctypes.data, ref,
                                           C.byref(md) if cfg >= 200 else None))
    if cfg >= 200:
        cv[0], cv[1] = 3, 1
    else:
        lib.hh_conv_config(cfg, cv)
    ks, s = cv[0], cv[1]
    ho, wo = (H // 2, W // 2) if s == 2 else (H, W)
    fl = 2.0 * B * ho * wo * cin * cout * ks * ks
    by = 2.0 * B * (H * W * cin + ho * wo * cout * (2 if res else 1))
    if st.any():
        t = st.astype(np.int64); nz = [(i, int(v - t[0])) for i, v in enumerate(t) if v]
        print("   stamps (cycles from kernel start; 0 start,1 loads issued,2 acc init,3+3c chunk c in LDS,4+3c chunk c MFMAs done,5+3c synced,15 end):", nz)
    tag = f"dma{cfg-200} maxdiff_vs_generic={md.value:.4g}" if cfg >= 200 else f"cfg{tuple(cv)}"
    print(f"{tag} B{B} {H}x{W} {cin}->{cout} res={res}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s  {by/ms.value/1e9:7.2f} TB/s(min traffic)")
