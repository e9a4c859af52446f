#!/bin/bash
# Round profile recipe (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
# Four separate rocprofv3 runs of the default bench workload: kernel trace + stats, then one PMC pass each for
# FETCH_SIZE and WRITE_SIZE (counters are never combined with other trace domains).  Outputs under gpurun_out/<tag>_*;
# tools/summarise_profile.py turns them into profiles/<tag>_summary.md, <tag>_bench_kernel_stats.csv, traffic_<tag>.json.
set -eo pipefail
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --no-cpu-baseline > $out/${tag}_stats.log 2>&1
echo "stats pass done"
# isolated kernel durations (one stream, no overlap between kernels): the figures bench.py's roofline probe must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_serial -- python3 bench.py --no-cpu-baseline --single-lane --sequential > $out/${tag}_serial.log 2>&1
echo "serial pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_write.log 2>&1
echo "write pass done"
# fp8 configuration (BASELINE.json configs[4]): serial kernel durations + HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_fp8_serial -- python3 bench.py --config fp8_w48_b64_640 --no-cpu-baseline --single-lane --steps 20 > $out/${tag}_fp8_serial.log 2>&1
echo "fp8 serial pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fp8_fetch -- python3 bench.py --config fp8_w48_b64_640 --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_fp8_fetch.log 2>&1
echo "fp8 fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_fp8_write -- python3 bench.py --config fp8_w48_b64_640 --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_fp8_write.log 2>&1
echo "fp8 write pass done"
