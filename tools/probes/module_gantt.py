"""Per-queue picture of one forward from a rocprofv3 kernel trace (tools/probes/timeline.sh writes one):
python3 tools/probes/module_gantt.py <kernel_trace.csv> [t0_us t1_us]  -- every kernel in [t0, t1) with start / end / queue, then
per queue the busy time in the window."""
import csv, sys, collections, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("stem_")]
i0, i1 = st[len(st) // 2], st[len(st) // 2 + 1]
fw = rows[i0:i1]
t0 = int(fw[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("conv_mfma_kernel", "conv")
    return n[:40]
qs = {}
busy = collections.Counter()
for r in fw:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if r["Kernel_Name"].startswith("stage_average"): break
    q = qs.setdefault(r["Queue_Id"], len(qs))
    if e < lo or s >= hi: continue
    busy[q] += min(e, hi) - max(s, lo)
    print(f"{s:8.1f} {e:8.1f} {e - s:6.1f}  q{q}  {'    ' * q}{short(r['Kernel_Name'])}  wg={r.get('Workgroup_Size','')} grid={r.get('Grid_Size','')}")
print("busy us per queue in the window:", {f"q{k}": round(v) for k, v in sorted(busy.items())})
