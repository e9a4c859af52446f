"""Same-box control for the "power wall" reading of the forward: what does a dense bf16 GEMM of the forward's FLOP
count hold on this box, at what board power and shader clock?  Run on the GPU box: python tools/probes/gemm_control.py [seconds]

Three operand shapes on random data (zeros clock higher: MI355X_MICROARCH.md, DVFS give-back):
  * 8192^3                       -- the library's best case: what the chip sustains when nothing but the matrix pipes limits it;
  * M = 32*128*128, N = 32,  K = 288   -- the 32-channel 3x3 convolution of branch 0 as a plain GEMM (im2col already done: an upper bound
  * M = 32*64*64,   N = 64,  K = 576      for any implicit-GEMM kernel of that layer, which also has to build the patches);
  * M = 32*32*32,   N = 128, K = 1152
each looped for `seconds` while rocm-smi is sampled for power and sclk, then the default forward loop the same way (bench.py's
net and batch, no decode).  Prints one line per workload: TFLOP/s, fraction of the 2.5 PFLOP/s dense peak, W, sclk MHz.
"""
import importlib
import os
import re
import subprocess
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
                sc = re.search(r"sclk clock level:\s*\d+:?\s*\(?([0-9.]+)Mhz", out)
                if pw and sc:
                    self.samples.append((float(pw.group(1)), float(sc.group(1))))
            except Exception:
                pass
            time.sleep(0.3)


def measure(name, flops_per_call, call):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    smi = Smi()
    smi.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    smi.stop = True
    smi.join()
    s = smi.samples[len(smi.samples) // 3:] or [(float("nan"), float("nan"))]  # the last two thirds: clocks and power have settled
    pw, sc = sum(x[0] for x in s) / len(s), sum(x[1] for x in s) / len(s)
    tf = flops_per_call * n / dt / 1e12
    print(f"{name:44s} {tf:8.1f} TFLOP/s = {tf / 2500:.3f} of peak   {pw:6.0f} W   sclk {sc:5.0f} MHz   ({n} calls, {dt / n * 1e3:.3f} ms each)", flush=True)
    return tf, pw, sc


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, (m, n, k) in {"gemm 8192 x 8192 x 8192": (8192, 8192, 8192), "gemm 524288 x 32 x 288 (32ch 3x3 @128^2)": (32 * 128 * 128, 32, 288),
                            "gemm 131072 x 64 x 576 (64ch 3x3 @64^2)": (32 * 64 * 64, 64, 576), "gemm 32768 x 128 x 1152 (128ch 3x3 @32^2)": (32 * 32 * 32, 128, 1152)}.items():
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        measure(name, 2.0 * m * n * k, lambda: torch.matmul(a, b, out=c))
        del a, b, c
    pkg = importlib.import_module("pytorch-human-pose_amd")
    net = pkg.HigherHRNet(17, 32)
    sd = {kk: torch.from_numpy(pkg.synth.synth_param(kk, v.shape, 0)) for kk, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net.to(dev).eval()
    x = torch.from_numpy(pkg.synth.synth_images(32, 512, 512, 0)).to(dev)
    with torch.no_grad():
        out = net.forward_raw(x)
        measure("HigherHRNet-W32 forward, B = 32 @ 512^2", 92.407e9 * 32, lambda: net.forward_raw(x, out))


if __name__ == "__main__":
    main()
