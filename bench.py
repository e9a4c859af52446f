#!/usr/bin/env python3
"""Benchmark of the north-star path: HigherHRNet-W32 forward + AE decode on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
  * hh_forward on a [32,3,512,512] fp32 batch (seeded N(0,1) images, seeded synthetic weights), then
  * hh_decode on a batch of constructed network-output maps with 10 people per image
    (random-weight outputs have no peaks, so decode gets constructed maps of the
    shapes the net emits: 17x128x128 + 17x256x256 heatmaps, 17x128x128 tags).
Multi-GPU: the path shards by image (no collective on the data path); every rank runs the same
per-GPU batch (weak scaling), timing is barrier + synchronize on both sides, max over ranks.

The two halves are issued back to back on ONE stream, so a step costs forward + decode ("reported
separately and summed"); `--overlap` puts the decode on a second stream beside the forward (serving: decode of batch i
runs beside the forward of batch i+1), which is a few percent faster and is NOT the headline.

Prints ONE JSON line (rank 0).  The K timed steps run as in production (eager launches, branch lanes on internal
streams).  `roofline` comes from probe steps AFTER the timed loop (not part of `value`): the kernels run one at a time on
one stream and every convolution launch carries a start / stop HIP event pair that the runtime fills from the dispatch
packet's own begin / end timestamps (hipExtLaunchKernelGGL, on the stream the kernel is launched on) -- the clock pair
rocprofv3's kernel trace reports, so `avg_launch_us` is directly comparable with the serial pass committed in
profiles/rNN_bench_kernel_stats.csv.  The dominant kernel instantiation is the one with the largest summed time.
`decode_roofline` = compulsory decode bytes over the time of hh_decode alone, against HBM peak.  `cpu_baseline` times
the CPU oracle (oracle/) on a bounded sample, on rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
PKG = "pytorch-human-pose_amd"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
MFMA_FP8_DENSE_PEAK_TFLOPS = 5000.0   # same table: "Peak FP8 MFMA ~5 PF dense"
REALISTIC_BF16_TFLOPS = 1385.0  # measured: a dense 8192^3 bf16 GEMM at the 1400 W board limit (profiles/r04_gemm_control.txt): the calibrated ceiling
# BASELINE.json configs a bench line can be quoted on: [1] (the headline, default) and [4] (the fp8 conv path)
CONFIGS = {
    "w32_b32_512": dict(C=32, dtype="bf16", batch=32, size=512, peak=MFMA_BF16_DENSE_PEAK_TFLOPS,
                        metric="images/sec (fwd+decode) HigherHRNet-W32 512px", name="HigherHRNet-W32 inference bf16"),
    # (not a BASELINE.json line: the fp8 workload's net, batch and size on the bf16 path, for the fp8 / bf16 ratio on one box)
    "bf16_w48_b64_640": dict(C=48, dtype="bf16", batch=64, size=640, peak=MFMA_BF16_DENSE_PEAK_TFLOPS,
                             metric="images/sec (fwd+decode) HigherHRNet-W48 bf16 640px", name="HigherHRNet-W48 inference bf16"),
    "fp8_w48_b64_640": dict(C=48, dtype="fp8", batch=64, size=640, peak=MFMA_FP8_DENSE_PEAK_TFLOPS,
                            metric="images/sec (fwd+decode) HigherHRNet-W48 fp8 640px", name="HigherHRNet-W48 inference fp8 (e4m3 MFMA conv path, EXPERIMENTAL: outputs 5-8 % rms from fp32 on seeded "
                                 "nets, no trained checkpoint to measure AP with; the decode half runs on constructed maps)"),
}


def collect_profile(pkg, net, peak=MFMA_BF16_DENSE_PEAK_TFLOPS):
    lib = pkg._lib.load()
    n = lib.hh_profile_count(net._h)
    """-> {cfg: {n, ms, kms, flops}}: per-launch times summed per kernel instantiation.  ms = HIP-event bracket on the
    launch stream; kms = the same launches timed by the kernel itself on the device wall clock (first workgroup start
    to last workgroup end), which is what rocprofv3's kernel trace reports."""
    per_cfg = {}
    cfg, flops, nbytes, ms, kms, name = C.c_int(), C.c_double(), C.c_double(), C.c_float(), C.c_float(), C.c_char_p()
    for i in range(n):
        pkg._lib.check(lib.hh_profile_get(net._h, i, C.byref(cfg), C.byref(flops), C.byref(nbytes), C.byref(ms), C.byref(kms), C.byref(name)))
        d = per_cfg.setdefault(cfg.value, {"n": 0, "ms": 0.0, "kms": 0.0, "flops": 0.0, "bytes": 0.0, "ceil_s": 0.0})
        d["n"] += 1
        d["ms"] += ms.value
        d["kms"] += kms.value if kms.value > 0 else ms.value
        ghz = C.c_double()
        pkg._lib.check(lib.hh_profile_clock(net._h, i, C.byref(ghz)))
        if ghz.value > 0:
            d.setdefault("ghz", []).append(ghz.value)
        d["flops"] += flops.value
        d["bytes"] += nbytes.value
        # roofline time of this launch: max(FLOPs / MFMA peak, algorithmic bytes / HBM peak)
        d["ceil_s"] += max(flops.value / (peak * 1e12), nbytes.value / (HBM_PEAK_GBS * 1e9))
    return per_cfg


def traffic_for(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed rocprofv3 PMC passes (profiles/traffic_rNN.json,
    written by tools/profile_round.sh + tools/summarise_profile.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "traffic_r*.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        return json.load(f)["bytes_per_launch"].get(kernel_name)


def cpu_baseline(pkg, sd, maps, fwd_images=4, dec_images=8, size=512):
    """The oracle (a port of the reference's path, kind="port") timed on this host, as BASELINE.md section 3 lays it out, on a bounded
    sample: forward fp32 at B = 1 and at B = `fwd_images` on min(32, nproc) torch threads (one warm-up, best of 3); decode of the
    constructed maps one image per thread -- the reference's decode is single-threaded Python, the C restatement is its stand-in --
    first alone (1 thread), then min(32, nproc) images side by side.  `value` = 1 / (best forward time per image + parallel decode
    time per image): what this host would sustain with every core busy on both halves."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import decode as orc
    from oracle import forward as ofw

    ncpu = os.cpu_count() or 1
    nthreads = min(32, ncpu)  # torch's CPU convs stop scaling (and regress) far below 128 threads
    torch.set_num_threads(nthreads)
    x = torch.from_numpy(pkg.synth.synth_images(fwd_images, size, size, 0))

    def best_of(fn, n=3):
        fn()  # warm-up
        best = float("inf")
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    with torch.no_grad():
        t_f1 = best_of(lambda: ofw.higher_hrnet(x[:1], sd, 17))
        t_fb = best_of(lambda: ofw.higher_hrnet(x, sd, 17), 2) / fwd_images
    orc.lib()

    def dec(i):
        hm_q, hm_h, tg = maps[i % len(maps)]
        return orc.decode(hm_q, hm_h, [tg], max_people=30, det_thr=0.05, tag_thr=0.5)

    t0 = time.perf_counter()
    for i in range(dec_images // 2):
        dec(i)
    t_d1 = (time.perf_counter() - t0) / (dec_images // 2)
    with ThreadPoolExecutor(nthreads) as ex:  # (the C oracle runs outside the GIL)
        t0 = time.perf_counter()
        list(ex.map(dec, range(nthreads)))
        t_dp = (time.perf_counter() - t0) / nthreads
    t_fwd = min(t_f1, t_fb)
    gf = 92.407 * (size / 512.0) ** 2
    return {
        "value": 1.0 / (t_fwd + t_dp),
        "unit": "images/sec",
        "cores": nthreads,
        "kind": "port",
        "sample": f"oracle fp32 forward {size}x{size} on {nthreads} torch threads: B=1 {t_f1 * 1e3:.0f} ms ({gf / t_f1:.0f} GFLOP/s), B={fwd_images} "
                  f"{t_fb * 1e3:.0f} ms/img (warm-up + best of 3 / 2); C oracle decode of the bench maps: 1 thread {t_d1 * 1e3:.0f} ms/img "
                  f"({dec_images // 2} images), {nthreads} images side by side {t_dp * 1e3:.1f} ms/img; value = 1 / (best forward + parallel decode); "
                  f"single-threaded decode instead: {1.0 / (t_fwd + t_d1):.2f} img/s; host has {ncpu} cpus",
    }


def timed_loop(step, warmup, steps, dist, sync, reduce_device):
    """The driver's timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device
    synchronize on both sides; returns the MAX over ranks of the elapsed seconds.  Shared by the inference bench, the
    training bench and the CPU rehearsal of the N > 1 launch contract (tests/test_distributed_cpu.py)."""
    last = None
    for _ in range(warmup):
        last = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, last


def headline(metric, world, per_gpu_batch, args, elapsed, dtype, config):
    """The one JSON line of the contract; `value` is the whole-job aggregate over all ranks."""
    return {
        "metric": metric, "value": round(world * per_gpu_batch * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": config,
    }


def rehearsal_cpu(args, dist, rank, world):
    """HH_BENCH_REHEARSAL=cpu: the N > 1 launch contract without a GPU (gloo): same process-group init, same timed_loop,
    same rank-0 line; a step is a sleep whose length depends on the rank, so the test can see that the MAX was taken."""
    elapsed, _ = timed_loop(lambda: time.sleep(0.002 * (rank + 1)), args.warmup, args.steps, dist, lambda: None, "cpu")
    if rank == 0:
        print(json.dumps(headline("images/sec (fwd+decode) HigherHRNet-W32 512px", world, args.batch, args, elapsed, "bf16",
                                  {"workload": "CPU rehearsal of the multi-rank launch contract (no GPU work)", "global_batch": world * args.batch,
                                   "parallelism": f"image-sharded replicas x{world}, no data-path collective"})), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def train_bench(args, pkg, dist, rank, world, dev):
    """Training step of configs[2] (keypoints/module.py:43-71): per-GPU batch `--batch` (32 x 8 GPUs = the global batch 256),
    synthetic images and targets, bf16 activations.  One step = forward (batch-statistics BN) + AEKeypointsLoss + backward +
    Adam; under torchrun the net is wrapped in DistributedDataParallel (gradient all-reduce on RCCL, overlapped with the
    backward by torch's bucketing).  Prints one JSON line like the inference bench (no roofline: the step is a mix of kernels)."""
    B, S, K = args.batch, 512, 17
    net = pkg.HigherHRNet(K, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    net = net.to(dev).train()
    model = net
    if dist is not None:  # base/model.py:36-48; experiments/keypoints/higher_hrnet_32.yaml:17 sets sync_batchnorm: false
        wrapper = pkg.keypoints.KeypointsModel(net)
        wrapper.to_DDP(dev.index, use_batchnorm=False)
        model = wrapper.net
    loss_fn = pkg.AEKeypointsLoss()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = torch.from_numpy(pkg.synth.synth_images(B, S, S, seed=rank)).to(dev)
    hms, masks, joints = pkg.synth.synth_train_targets(B, K, S, args.people, seed=rank)
    hms = [torch.from_numpy(h).to(dev) for h in hms]
    masks = [torch.from_numpy(m).to(dev) for m in masks]

    def step():
        ph, pt = model(x)
        hl, push, pull = loss_fn.calculate_loss(ph, pt, hms, masks, joints)
        loss = hl[0] + hl[1] + push[0] + pull[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    # the step runs on a highest-priority stream (dependent launches follow each other faster there, see bench())
    prio = int(os.environ.get("HH_STREAM_PRIORITY", str(torch.cuda.Stream.priority_range()[1])))
    train_stream = torch.cuda.Stream(dev, priority=prio)
    train_stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(train_stream):
        args.warmup = max(args.warmup, 1)
        elapsed, loss = timed_loop(step, args.warmup, args.steps, dist, torch.cuda.synchronize, dev)
    if rank == 0:
        print(json.dumps(headline(
            "images/sec (training step) HigherHRNet-W32 512px", world, B, args, elapsed, "bf16",
            {"workload": f"HigherHRNet-W32 training step, batch {B} @ 512x512 per GPU: forward (train-mode BN) + AE loss + "
                         f"backward + Adam, {args.people} people/image", "global_batch": world * B,
             "parallelism": f"DistributedDataParallel x{world} (gradient all-reduce on RCCL, BatchNorm per rank as in the reference's experiment file)" if world > 1 else "single GPU",
             "final_loss": round(float(loss.item()), 5),
             "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)})), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _visible_gpus():
    """GPUs this process would see, counted without a HIP call (torch.cuda.device_count() may go through hipGetDeviceCount on a ROCm
    build and leave an initialised runtime in the parent): the KFD topology's nodes with SIMDs, cut by HIP_/ROCR_VISIBLE_DEVICES."""
    n = 0
    try:
        root = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None if os.path.exists("/dev/kfd") else 0  # a driver without a readable topology: let the ranks find out; no driver: no GPUs
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the N-rank job as a CHILD `torch.distributed.run` (rendezvous on 127.0.0.1,
    a free port) with this command line, and exit with its code.  It must stay a child process and never become an exec: a process
    that has touched the GPU must not replace itself on this pool.  The parent makes no HIP call at all (the device count comes from
    sysfs), so the ranks are the first GPU users.  A rendezvous port that was taken between the probe and the launch: try again."""
    import socket
    import subprocess

    if not os.environ.get("HH_BENCH_REHEARSAL"):
        have = _visible_gpus()
        if have is not None and have < n:
            sys.exit(f"bench.py: --gpus {n} but this node shows {have} GPU(s)")
    rc = 1
    for _attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(proc.stderr)
        rc = proc.returncode
        if rc == 0 or "Address already in use" not in proc.stderr:
            break
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default: the config's)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="w32_b32_512",
                    help="BASELINE.json workload: w32_b32_512 = configs[1] (headline), fp8_w48_b64_640 = configs[4]; bf16_w48_b64_640 = that workload on the bf16 path")
    ap.add_argument("--people", type=int, default=10)
    ap.add_argument("--dense-people", type=int, default=27,
                    help="people per image of the second, dense set of constructed maps whose decode time is reported beside the headline "
                         "(config.decode_dense_ms; 0 = skip): the 10-people maps of SURVEY.md section 8d understate what a crowded image costs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch probe steps behind the timed loop")
    ap.add_argument("--probe-steps", type=int, default=3, help="serial probe steps behind the timed loop (roofline line)")
    ap.add_argument("--overlap", action="store_true",
                    help="issue the decode on a second stream beside the forward instead of behind it (not the headline)")
    ap.add_argument("--sequential", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--chained", action="store_true",
                    help="the decode consumes the forward's OWN outputs: the net gets the seeded pass-through weights (dense random "
                         "channels + reserved channels that carry constructed heatmaps / tags encoded in the input images to the "
                         "outputs, pkg.synth.synth_passthrough_*), so its maps hold --people people per image")
    ap.add_argument("--single-lane", action="store_true",
                    help="no internal branch streams: kernels run one after another (what the roofline probe and the "
                         "isolated-kernel rocprofv3 pass measure)")
    ap.add_argument("--train", action="store_true",
                    help="time the training step of BASELINE.json configs[2] instead (forward with train-mode BN + AE loss + "
                         "backward + Adam, DistributedDataParallel over RCCL when launched on several GPUs); not the headline metric")
    args = ap.parse_args()

    cfgd = CONFIGS[args.config]
    if args.batch is None:
        args.batch = cfgd["batch"]
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves -- before this process has made any GPU call -- exactly as
        # the driver would (one process per GPU under torch.distributed.run), and leave with the launcher's exit code
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # never print a line whose n_gpus is not what the command line asked for
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                 f"or run plain `python bench.py --gpus {args.gpus}` and let it start the ranks")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HH_BENCH_REHEARSAL") == "cpu":
            dist.init_process_group(backend="gloo")
            return rehearsal_cpu(args, dist, rank, world)
        if os.environ.get("HH_BENCH_REHEARSAL"):
            # rehearsal of the N > 1 code path on a ONE-GPU box (every rank on cuda:0, gloo): checks the launch contract
            # (barriers, max over ranks, rank-0 line), not performance - the driver's multi-GPU runs use RCCL below
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    pkg = importlib.import_module(PKG)

    B, H, W, K = args.batch, cfgd["size"], cfgd["size"], 17
    if args.train:
        return train_bench(args, pkg, dist, rank, world, dev)
    net = pkg.HigherHRNet(K, cfgd["C"], dtype=cfgd["dtype"])
    if args.chained:
        sd = {k: torch.from_numpy(v) for k, v in pkg.synth.synth_passthrough_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, K, 0).items()}
        images = torch.from_numpy(pkg.synth.synth_passthrough_images(B, H // 4, W // 4, args.people, K, seed=rank)[0]).to(dev)
    else:
        sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()}
        images = torch.from_numpy(pkg.synth.synth_images(B, H, W, seed=rank)).to(dev)
    net.load_state_dict(sd)
    net.to(dev).eval()
    if cfgd["dtype"] == "fp8":  # activation scales from a calibration batch of the same distribution (other seed)
        net.calibrate(torch.from_numpy(pkg.synth.synth_images(8, H, W, seed=4242)).to(dev))
    uniq = [pkg.synth.synth_decode_maps(K, H // 4, W // 4, args.people, seed=1000 + rank * 8 + i)[:3] for i in range(8)]
    uniq = [(a, b, t[0]) for a, b, t in uniq]
    hm_q = torch.from_numpy(np.stack([uniq[i % 8][0] for i in range(B)])).to(dev)
    hm_h = torch.from_numpy(np.stack([uniq[i % 8][1] for i in range(B)])).to(dev)
    tags = torch.from_numpy(np.stack([uniq[i % 8][2] for i in range(B)])).to(dev)
    parser = pkg.MPPEHeatmapParser(K, 30, 0.05, 0.5)
    lib = pkg._lib.load()
    dense = None
    if args.dense_people > 0:
        du = [pkg.synth.synth_decode_maps(K, H // 4, W // 4, args.dense_people, seed=5000 + rank * 8 + i)[:3] for i in range(8)]
        dense = tuple(torch.from_numpy(np.stack([(du[i % 8][j] if j < 2 else du[i % 8][2][0]) for i in range(B)])).to(dev) for j in range(3))

    if args.single_lane:
        lib.hh_set_multi_lane(net._h, 0)
    # Highest-priority streams: dependent launches on a high-priority queue follow each other faster on this stack (forward
    # alone 5.25 -> 4.93 ms, step 5.85 -> 5.53 ms; round-2 A/B).  HH_STREAM_PRIORITY=0 for A/B runs.
    prio = int(os.environ.get("HH_STREAM_PRIORITY", str(torch.cuda.Stream.priority_range()[1])))
    side = torch.cuda.Stream(dev, priority=prio)   # forward stream (the engine forks its branch lanes from it)
    side2 = torch.cuda.Stream(dev, priority=prio)  # decode stream
    outs = (torch.empty(B, 2 * K, H // 4, W // 4, device=dev), torch.empty(B, K, H // 2, W // 2, device=dev))

    def step(isolate=False):
        # forward, then decode behind it on the same stream: a step costs their sum.  --overlap: the two
        # halves have no data dependency (the decode consumes constructed maps), so the decode may run on a second stream.
        out = net.forward_raw(images, outs)
        if args.chained:  # decode what the forward just wrote: heatmaps = init[:, :K] / deconv, tags = init[:, K:]
            dec = parser.decode_batch_device(out[0][:, :K], out[1], [out[0][:, K:]], adjust=True, refine=True)
        elif args.overlap and not isolate:
            with torch.cuda.stream(side2):
                dec = parser.decode_batch_device(hm_q, hm_h, [tags], adjust=True, refine=True)
        else:
            dec = parser.decode_batch_device(hm_q, hm_h, [tags], adjust=True, refine=True)
        return out, dec

    stream = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        profile = not args.no_profile
        net.use_graph = True
        elapsed, (out, dec) = timed_loop(step, args.warmup, args.steps, dist, torch.cuda.synchronize, dev)
        per_cfg = {}
        if profile:
            # probe steps, outside the timed region: conv launches one at a time on one stream, each with its own start / stop
            # event pair (hipExtLaunchKernelGGL), the decode behind them on the same stream
            lib.hh_profile_enable(net._h, 1)
            for _ in range(max(1, args.probe_steps)):
                step(isolate=True)
            side.synchronize()
            per_cfg = collect_profile(pkg, net, cfgd["peak"])
            # one more step with the in-kernel device-clock stamps (a side figure; its atomics lengthen the launches, so it
            # is never mixed with the event timing above)
            lib.hh_profile_enable(net._h, 2)
            step(isolate=True)
            side.synchronize()
            for c, d in collect_profile(pkg, net, cfgd["peak"]).items():
                if c in per_cfg:
                    per_cfg[c]["kms"] = d["kms"] * per_cfg[c]["n"] / d["n"]
                    if d.get("ghz"):
                        per_cfg[c]["ghz"] = d["ghz"]
            lib.hh_profile_enable(net._h, 0)
        # split of the step (not part of the timed region): forward alone / decode alone, graph replay
        net.use_graph = True
        for fn in (lambda: net.forward_raw(images, outs), lambda: parser.decode_batch_device(hm_q, hm_h, [tags])):
            fn()
        parts = []
        for fn, reps in ((lambda: net.forward_raw(images, outs), 10), (lambda: parser.decode_batch_device(hm_q, hm_h, [tags]), 40)):
            side.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):  # (the decode is 0.3 ms: five calls were a 1.5 ms window and read 0.28 or 0.37 from run to run)
                fn()
            side.synchronize()
            parts.append((time.perf_counter() - t1) / reps)
        dense_ms = dense_people = None
        if dense is not None:  # a crowded batch through the same decoder (not part of `value`)
            fn = lambda: parser.decode_batch_device(dense[0], dense[1], [dense[2]], adjust=True, refine=True)  # noqa: E731
            dd = fn()
            side.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                dd = fn()
            side.synchronize()
            dense_ms = (time.perf_counter() - t1) / 20 * 1e3
            dense_people = int(dd[2].sum().item())
    del stream
    num_people = int(dec[2].sum().item())
    assert not bool(dec[3].any().item()), "decode flagged an image (solver guard / no-group fallback) on the bench maps"

    if rank == 0:
        line = headline(cfgd["metric"], world, B, args, elapsed, cfgd["dtype"], {
                "workload": f"{cfgd['name']}, batch {B} @ {H}x{W} per GPU (" + (
                                f"CHAINED: forward of the pass-through net on images that encode {args.people} constructed people each, "
                                "AE decode of the forward's own outputs" if args.chained else
                                f"conv fwd on N(0,1) images + AE decode on constructed maps, {args.people} people/image") +
                            ", det_thr 0.05, tag_thr 0.5, adjust+refine)",
                "global_batch": world * B,
                "parallelism": f"image-sharded replicas x{world}, no data-path collective",
                "people_decoded_per_batch": num_people,
                "forward_ms": round(parts[0] * 1e3, 3),
                "decode_ms": round(parts[1] * 1e3, 3),
                "decode_dense_ms": None if dense_ms is None else round(dense_ms, 3),
                "decode_dense_people_per_batch": dense_people,
                "forward_tflops": round(net.forward_flops(B, H, W) / parts[0] / 1e12, 1),
                "streams": "forward (+3 internal branch lanes) and decode on two streams" if args.overlap else
                           "forward (+3 internal branch lanes), then decode, back to back on one stream: step = forward + decode",
                "timed_region": "eager launches as in production; the per-launch roofline probe runs in separate steps behind the timed loop",
        })
        if per_cfg:
            dom = max(per_cfg, key=lambda c: per_cfg[c]["ms"])
            d = per_cfg[dom]
            def kernel_name(c):
                nm = names.get(c, str(c))
                cv = (C.c_int * 7)()
                if lib.hh_conv_config(c, cv) == 0:
                    nm = ("conv_fp8_kernel" if c >= 1000 else "conv_mfma_kernel") + "<KS=%d,S=%d,KC=%d,NT=%d,WC=%d,PT=%d,TW=%d%s>" % (
                        tuple(cv) + (",DB=1" if lib.hh_conv_config_double_buffered(c) == 1 else "",))
                return nm
            names = {100: "bb_fused_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=32)" if os.environ.get("HH_BB32") == "tile" else
                          "bbpc_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=32, producer/consumer waves)",
                     105: "bb_fp8_kernel (fused e4m3 BasicBlock: conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU)",
                     103: "bb64_fused_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=64)",
                     102: "stem_conv_kernel (fp32 NCHW -> conv3x3 s2 3->64 + BN + ReLU -> bf16 NHWC)",
                     106: "stem_fused_kernel (fp32 NCHW -> conv3x3 s2 3->64 + BN + ReLU -> conv3x3 s2 64->64 + BN + ReLU -> bf16 NHWC)",
                     101: "junction_kernel (stage-0 conv3 1x1 [+downsample] + residual + ReLU + next conv1 1x1 + ReLU)"}
            kname = kernel_name(dom)
            # Per-launch duration = the dispatch packet's begin .. end timestamps, delivered through the start / stop events of
            # hipExtLaunchKernelGGL on the launch stream: the quantity rocprofv3's kernel trace reports (serial pass in
            # profiles/rNN_bench_kernel_stats.csv).  The kernel's own first-workgroup-start .. last-workgroup-end on the device
            # wall clock is reported beside it (shorter: no dispatch ramp, no end-of-kernel write-back).
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in per_cfg.values())
            line["roofline"] = {
                "bound": "mfma",
                "achieved": round(achieved, 2),
                "peak": cfgd["peak"],
                "unit": "TFLOP/s",
                "frac": round(achieved / cfgd["peak"], 4),
                "traffic": traffic_for(kname),
                "kernel": kname,
                "launches": d["n"],
                "avg_launch_us": round(d["ms"] / d["n"] * 1e3, 2),
                "avg_launch_us_device_clock": round(d["kms"] / d["n"] * 1e3, 2),
                "achieved_device_clock": round(d["flops"] / (d["kms"] * 1e-3) / 1e12, 2),
                "timing": "HIP start/stop events of each launch (hipExtLaunchKernelGGL: dispatch begin/end timestamps, what rocprofv3's "
                          "kernel trace reports), probe steps behind the timed loop, kernels serialised on one stream",
                "avg_launch_gflop": round(d["flops"] / d["n"] / 1e9, 3),
                # the same launches against the HBM side of the roofline: algorithmic bytes (input + output + residual +
                # weights, once each) over the same durations, and the two-sided ceiling min(MFMA peak, AI x HBM peak)
                "avg_launch_mbytes": round(d["bytes"] / d["n"] / 1e6, 2),
                "achieved_hbm_gbs": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                "hbm_frac": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "roofline_ceiling_tflops": round(d["flops"] / d["ceil_s"] / 1e12, 1),
                # what the chip holds on a plain dense bf16 GEMM at its board power limit (hipBLASLt 8192^3 on random data, same box as
                # the forward: 1385 TFLOP/s at 1390 W and sclk 1.88 GHz, tools/probes/gemm_control.py, profiles/r04_gemm_control.txt)
                "realistic_ceiling_tflops": REALISTIC_BF16_TFLOPS if cfgd["dtype"] == "bf16" else None,
                "frac_of_realistic_ceiling": round(achieved / REALISTIC_BF16_TFLOPS, 4) if cfgd["dtype"] == "bf16" else None,
                "frac_of_ceiling": round(d["ceil_s"] / (d["ms"] * 1e-3), 4),
                # the core clock workgroup 0 of these launches ran at (s_memtime / s_memrealtime deltas in the extra, stamped probe step): the
                # chip holds it below its 2.4 GHz under this load, and `peak` is quoted at 2.4 GHz
                "in_kernel_clock_ghz": round(sum(d["ghz"]) / len(d["ghz"]), 3) if d.get("ghz") else None,
                "frac_of_peak_at_that_clock": round(achieved / (cfgd["peak"] * (sum(d["ghz"]) / len(d["ghz"])) / 2.4), 4) if d.get("ghz") else None,
                "share_of_conv_time": round(d["ms"] / total_ms, 3),
                # the next kernels by summed time (the first two are close: which one leads differs from box to box)
                "runners_up": [{"kernel": kernel_name(c), "launches": v["n"], "avg_launch_us": round(v["ms"] / v["n"] * 1e3, 2),
                                "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / cfgd["peak"], 4), "share_of_conv_time": round(v["ms"] / total_ms, 3)}
                               for c, v in sorted(per_cfg.items(), key=lambda kv: -kv[1]["ms"])[1:(None if os.environ.get("HH_BENCH_ALL_KERNELS") else 4)]],
                "all_conv_tflops": round(sum(v["flops"] for v in per_cfg.values()) / (total_ms * 1e-3) / 1e12, 2),
            }
        # decode half against the HBM roofline: compulsory bytes (read every network output once) over the
        # time of one hh_decode call alone; the kernels are latency/VALU bound by design, so this is small
        dec_bytes = 4.0 * K * ((H // 4) * (W // 4) * 2 + (H // 2) * (W // 2)) * B  # 6,684,672 B per 512x512 image
        line["decode_roofline"] = {
            "bound": "hbm", "achieved": round(dec_bytes / parts[1] / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(dec_bytes / parts[1] / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic_for("hh_decode (all kernels of one call)"),
            "algorithmic_bytes_per_call": dec_bytes, "ms_per_call": round(parts[1] * 1e3, 3),
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only: the other ranks would idle at the final barrier
            line["cpu_baseline"] = cpu_baseline(pkg, sd, uniq, fwd_images=4 if H <= 512 else 2, size=H)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
