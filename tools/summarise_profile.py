"""Condense the rocprofv3 outputs of tools/profile_round.sh into the tracked files under profiles/.

    python3 tools/summarise_profile.py r01

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB per dispatch, collected
in separate passes; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are doubled.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True), key=os.path.getmtime)  # newest run wins
    if not hits:
        sys.exit(f"missing {pattern}")
    return hits[-1]


def pmc(name, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(one(f"{tag}_{name}/**/*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}  # bytes per launch


stats = list(csv.DictReader(open(one(f"{tag}_stats/**/*kernel_stats.csv"))))
serial = {r["Name"]: r for r in csv.DictReader(open(one(f"{tag}_serial/**/*kernel_stats.csv")))}
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
with open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev",
                "SerialAverageNs", "SerialPercentage", "hbm_read_bytes_per_launch(2xFETCH_SIZE)", "hbm_write_bytes_per_launch"])
    for r in stats:
        n = r["Name"]
        sr = serial.get(n, {})
        w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"],
                    sr.get("AverageNs", ""), sr.get("Percentage", ""), round(2 * fetch.get(n, 0.0)), round(write.get(n, 0.0))])


def short(n):
    return n if len(n) <= 84 else n[:81] + "..."


lines = [f"# {tag} rocprofv3 summaries (bench.py default workload: HigherHRNet-W32, batch 32 @ 512x512, 1x MI355X)", "",
         "Recipe: `bash tools/profile_round.sh " + tag + "` = `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py "
         "--no-cpu-baseline`, then separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (`--steps 3 --warmup 1`).",
         "`avg us (overlapped)` is the default bench (4 internal streams + the decode stream: kernels share the chip, so each one's "
         "wall duration is longer than its isolated duration); `avg us (serial)` is `bench.py --single-lane --sequential` (one kernel "
         "at a time) and is the per-launch duration that bench.py's `roofline` probe reports from HIP events.",
         "HBM read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB, averaged per launch.",
         f"Full table: {tag}_bench_kernel_stats.csv.  `__amd_rocclr_copyBuffer`/`fillBuffer` rows are the one-off weight uploads "
         "and workspace clears of model setup, not part of a step.", "",
         "| kernel | calls | avg us (overlapped) | avg us (serial) | % time (serial) | HBM read MB/launch | HBM write MB/launch |",
         "|---|---|---|---|---|---|---|"]
for r in stats:
    if float(r["Percentage"]) < 0.2:
        continue
    n = r["Name"]
    sr = serial.get(n)
    lines.append(f"| `{short(n)}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(sr['AverageNs']) / 1e3 if sr else float('nan'):.1f} | "
                 f"{float(sr['Percentage']) if sr else float('nan'):.2f} | "
                 f"{2 * fetch.get(n, 0.0) / 1e6:.1f} | {write.get(n, 0.0) / 1e6:.1f} |")
# SQ counter pass (tools/pmc_pass.sh): MFMA pipe occupancy and where the waves' cycles go
sq_files = sorted(glob.glob(os.path.join(out, f"{tag}_sq/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)
if sq_files:
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
    with open(sq_files[-1]) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += r["Counter_Name"] == "SQ_WAVE_CYCLES"
    lines += ["", "## SQ counters (`bash tools/pmc_pass.sh " + tag + "`: serial bench, `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
              "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE`)", "",
              "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (serial duration x 2.4 GHz x 1024 SIMDs); a 32x32x16 bf16 MFMA holds the pipe "
              "32 cycles, so busy cycles = 32 x MFMAs issued (halo recompute and identity-residual MFMAs included).  WAIT_ANY = parked on "
              "s_waitcnt/barrier, WAIT_INST = issue stall (mostly the matrix pipe held by the other wave of the SIMD), ACTIVE = issuing.", "",
              "| kernel | MFMA busy cycles / launch | MFMA util (at 2.4 GHz) | wave cycles: wait_any / wait_inst / active % | LDS bank-conflict cycles / launch |",
              "|---|---|---|---|---|"]
    rows = []
    for n, v in acc.items():
        k = cnt[n] or 1
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and n in serial:
            dur = float(serial[n]["AverageNs"]) * 1e-9
            wc = v["SQ_WAVE_CYCLES"] or 1.0
            rows.append((float(serial[n]["Percentage"]), f"| `{short(n)}` | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / k:.3g} | "
                         f"{v['SQ_VALU_MFMA_BUSY_CYCLES'] / k / (dur * 2.4e9 * 1024):.3f} | {100 * v['SQ_WAIT_ANY'] / wc:.0f} / "
                         f"{100 * v['SQ_WAIT_INST_ANY'] / wc:.0f} / {100 * v['SQ_ACTIVE_INST_ANY'] / wc:.0f} | {v['SQ_LDS_BANK_CONFLICT'] / k:.3g} |"))
    lines += [r for _, r in sorted(rows, reverse=True)[:12]]
for which in ("stats", "serial"):
    for ln in open(os.path.join(out, f"{tag}_{which}.log"), errors="replace"):
        if ln.startswith('{"metric"'):
            lines += ["", f"bench.py line of the traced `{which}` run (profiler attached, so slower than an untraced run):", "", "```", ln.strip(), "```"]
# fp8 configuration (BASELINE.json configs[4]): serial kernel-trace pass + FETCH/WRITE passes of bench.py --config fp8_w48_b64_640
fp8_serial = sorted(glob.glob(os.path.join(out, f"{tag}_fp8_serial/**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
fp8_traffic = {}
if fp8_serial:
    rows = list(csv.DictReader(open(fp8_serial[-1])))
    try:
        ffetch, fwrite = pmc("fp8_fetch", "FETCH_SIZE"), pmc("fp8_write", "WRITE_SIZE")
    except SystemExit:
        ffetch, fwrite = {}, {}
    with open(os.path.join(prof, f"{tag}_fp8_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "SerialAverageNs", "SerialPercentage", "hbm_read_bytes_per_launch(2xFETCH_SIZE)", "hbm_write_bytes_per_launch"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], round(2 * ffetch.get(r["Name"], 0.0)), round(fwrite.get(r["Name"], 0.0))])
    lines += ["", "## fp8 configuration (`bench.py --config fp8_w48_b64_640 --single-lane`: HigherHRNet-W48, batch 64 @ 640x640, e4m3 MFMA conv path)", "",
              f"Serial kernel-trace pass (one kernel at a time), full table {tag}_fp8_kernel_stats.csv:", "",
              "| kernel | calls | avg us (serial) | % time | HBM read MB/launch | HBM write MB/launch |", "|---|---|---|---|---|---|"]
    for r in rows[:14]:
        n = r["Name"]
        lines.append(f"| `{short(n)}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | {2 * ffetch.get(n, 0.0) / 1e6:.1f} | {fwrite.get(n, 0.0) / 1e6:.1f} |")
    for ln in open(os.path.join(out, f"{tag}_fp8_serial.log"), errors="replace"):
        if ln.startswith('{"metric"'):
            lines += ["", "bench.py line of that traced run:", "", "```", ln.strip(), "```"]
    for n in set(ffetch) | set(fwrite):
        m = re.search(r"conv_fp8_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
        if m:
            ks, s_, kc, nt, pt, tw = m.groups()
            fp8_traffic[f"conv_fp8_kernel<KS={ks},S={s_},KC={kc},NT={nt},WC=1,PT={pt},TW={tw}>"] = round(2 * ffetch.get(n, 0.0) + fwrite.get(n, 0.0), -5)
        elif "bb_fp8_kernel" in n:  # (the bench names the fused e4m3 block without its template arguments: the larger of its instantiations)
            key = "bb_fp8_kernel (fused e4m3 BasicBlock: conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU)"
            fp8_traffic[key] = max(fp8_traffic.get(key, 0.0), round(2 * ffetch.get(n, 0.0) + fwrite.get(n, 0.0), -5))
train_csv = os.path.join(prof, f"{tag}_train_step_kernel_stats.csv")
if os.path.exists(train_csv):  # kept from `rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py 32 5` (7 steps traced)
    trows = list(csv.DictReader(open(train_csv)))
    lines += ["", "## Training step (`rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py 32 5`: HigherHRNet-W32, batch 32 @ 512x512, "
              "forward with train-mode BN + AE loss + backward + Adam; 2 warm-up + 5 timed steps traced, the one-off Adam state fills and weight uploads included)", "",
              f"Top kernels of {tag}_train_step_kernel_stats.csv (step wall time 60 ms untraced):", "",
              "| kernel | calls | avg us | % of kernel time |", "|---|---|---|---|"]
    lines += [f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |" for r in trows[:14]]
probes = [f for f in (f"{tag}_conv_probe.txt", f"{tag}_nms_probe.txt", f"{tag}_dispatch_probe.txt") if os.path.exists(os.path.join(prof, f))]
if probes:  # probe outputs (tools/probes/build.sh; kernel sources compiled with phase stamps)
    lines += ["", "## Kernel-internal probes", "", "Phase stamps inside the kernels (s_memtime per workgroup), start skew and inter-launch gaps: " +
              ", ".join(f"`{f}`" for f in probes) + " (built by `tools/probes/build.sh`, see DESIGN.md section 6)."]
open(os.path.join(prof, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")

# per-launch HBM bytes keyed the way bench.py names kernels
traffic = {}
for n in set(fetch) | set(write):
    total = 2 * fetch.get(n, 0.0) + write.get(n, 0.0)
    m = re.search(r"conv_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        traffic["conv_mfma_kernel<KS=%s,S=%s,KC=%s,NT=%s,WC=%s,PT=%s,TW=%s>" % m.groups()] = round(total, -5)
    elif "junction_kernel" in n:
        key = "junction_kernel (stage-0 conv3 1x1 [+downsample] + residual + ReLU + next conv1 1x1 + ReLU)"
        traffic[key] = max(traffic.get(key, 0.0), round(total, -5))
    elif n.startswith("bb_fused_kernel"):
        traffic["bb_fused_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=32)"] = round(total, -5)
    elif n.startswith("bbpc_kernel"):
        traffic["bbpc_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=32, producer/consumer waves)"] = round(total, -5)
    elif n.startswith("bb64_fused_kernel"):
        traffic["bb64_fused_kernel (conv3x3+BN+ReLU+conv3x3+BN+residual+ReLU, C=64)"] = round(total, -5)
DECODE = ("stage_average", "nms_classify", "nms_tile_topk", "peaks_region", "fallback_top1", "topk_merge", "match_kernel", "adjust_scores", "refine_", "tag_bounds")
traffic.update(fp8_traffic)
traffic["hh_decode (all kernels of one call)"] = round(sum(2 * fetch.get(n, 0.0) + write.get(n, 0.0) for n in set(fetch) | set(write)
                                                           if n.startswith(DECODE)), -5)
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 bench.py --no-cpu-baseline --steps 3 --warmup 1` "
                     f"(profiles/{tag}_summary.md); read = 2*FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB, averaged per launch",
           "bytes_per_launch": traffic}, open(os.path.join(prof, f"traffic_{tag}.json"), "w"), indent=1)
print("\n".join(lines[:20]))
