"""Soak of the suite order in which round 2 saw its one unexplained failure: in ONE process, repeat the forward tests of
tests/test_gpu_parity.py in file order up to and including test_forward_full_size_samples_and_batch_consistency (the test
functions themselves are called, so every handle / switch / shape of the suite is built, used and dropped exactly as there),
and stop at the first assertion with its full text.

    python tools/probes/soak_suite_order.py [iterations=200] [minutes=12]

Seeded parameters are cached between iterations (the suite regenerates them per net; that is host time, not what is soaked).
"""
import functools
import importlib
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0

pkg = importlib.import_module("pytorch-human-pose_amd")
pkg.synth.synth_param = functools.lru_cache(maxsize=None)(pkg.synth.synth_param)
pkg.synth.synth_images = functools.lru_cache(maxsize=None)(pkg.synth.synth_images)
t = importlib.import_module("test_gpu_parity")
golden = np.load(os.path.join(REPO, "tests", "golden", "net_forward.npz"))
golden = {k: golden[k] for k in golden.files}


class G(dict):
    files = property(lambda self: list(self.keys()))


golden = G(golden)
params = [("w32_128", 32, 2, 128, 128, 1), ("w32_96x160", 32, 1, 96, 160, 2), ("w48_64", 48, 1, 64, 64, 3)]
order = [
    lambda: t.test_forward_with_taps_vs_reference_golden(pkg, golden),
    *[functools.partial(t.test_forward_outputs_vs_reference_golden, pkg, golden, *p) for p in params],
    lambda: t.test_fused_32_channel_block_both_forms(pkg, golden),
    lambda: t.test_fused_stem_matches_two_launches(pkg, golden),
    lambda: t.test_schedule_and_fusion_switches(pkg, golden),
    lambda: t.test_forward_reads_no_unwritten_workspace(pkg),
    lambda: t.test_forward_full_size_samples_and_batch_consistency(pkg, golden),
]
t0 = time.time()
done = 0
for it in range(iters):
    for k, fn in enumerate(order):
        try:
            fn()
        except Exception:  # keep the text this time
            print(f"iteration {it}, step {k}: FAILED\n{traceback.format_exc()}", flush=True)
            sys.exit(1)
    done += 1
    if it % 10 == 0:
        print(f"iteration {it} clean, {time.time() - t0:.0f} s", flush=True)
    if time.time() - t0 > minutes * 60:
        break
print(f"soak: {done} iterations of the suite order clean in {time.time() - t0:.0f} s", flush=True)
