"""A/B of the two fused 32-channel BasicBlock kernels (tile form vs producer/consumer form) through hh_debug_bb_compare."""
import ctypes as C, importlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
LONG = int(os.environ.get("BB_LONG", "0"))  # 1: ~1 s of back-to-back launches per kernel and size (what the in-kernel clock stamps of a -DHH_STAMP build need)
for (B, H, W, it) in [(1, 14, 32, 2), (2, 30, 44, 2), (3, 61, 77, 2), (32, 128, 128, 30000 if LONG else 30), (32, 256, 256, 10000 if LONG else 10)]:
    md, m0, m1 = C.c_float(), C.c_float(), C.c_float()
    pkg._lib.check(lib.hh_debug_bb_compare(B, H, W, it, C.byref(md), C.byref(m0), C.byref(m1)))
    fl = 2 * 2.0 * B * H * W * 32 * 32 * 9
    print(f"B{B} {H}x{W}: max|diff| {md.value:.5f}   tile form {m0.value*1e3:.1f} us ({fl/m0.value/1e9:.0f} TF/s)   "
          f"producer/consumer {m1.value*1e3:.1f} us ({fl/m1.value/1e9:.0f} TF/s)", flush=True)
