"""Fused BasicBlock micro-benchmark (+ phase stamps when csrc is built with EXTRA=-DHH_STAMP)."""
import ctypes as C, importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("pytorch-human-pose_amd")
lib = pkg._lib.load()
for (B, H, W) in [(32, 128, 128), (32, 256, 256)]:
    ms = C.c_float(); st = np.zeros(64, np.uint64)
    pkg._lib.check(lib.hh_debug_bb_bench(B, H, W, 30, C.byref(ms), st.ctypes.data))
    fl = 2 * 2.0 * B * H * W * 32 * 32 * 9
    print(f"bb_fused B{B} {H}x{W}: {ms.value*1e3:.1f} us  {fl/ms.value/1e9:.1f} TF/s  {2.0*B*H*W*64/ms.value/1e9:.2f} TB/s(min)")
    st = st.reshape(8, 8).astype(np.int64)
    if st.any():
        for t in range(min(4, 8)):
            if st[t].any():
                d = np.diff(st[t])
                print("  tile", t, "phases[issue+conv1, epi1, sync, conv2, sync, stage+sync, copyout, sync]:", d.tolist(), "total", int(st[t][7] - st[t][0]))
