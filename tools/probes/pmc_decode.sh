#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $out/dpmc1 -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/dpmc1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $out/dpmc2 -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/dpmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("dpmc1", "dpmc2"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True))
    if not fs: print(d, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:28]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k in agg:
        if any(t in k for t in ("nms", "refine_arg", "tag_bounds", "stage_av", "topk_merge")):
            print(d, k, n[k], {c: f"{v / n[k]:.3g}" for c, v in agg[k].items()})
PY
