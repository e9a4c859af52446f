"""Timeline statistics of one forward from a rocprofv3 kernel trace: time with 0 / 1 / 2 / 3 / 4 kernels running, the bubbles."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("stem_")]
i0, i1 = st[len(st) // 2], st[len(st) // 2 + 1]
fw = rows[i0:i1]
t0 = int(fw[0]["Start_Timestamp"])
ev = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Kernel_Name"]) for r in fw]
dec = next((s for s, e, q, n in ev if n.startswith("stage_average")), ev[-1][1])
ev = [x for x in ev if x[0] < dec]
end = max(e for s, e, q, n in ev)
pts = sorted([(s, 1) for s, e, q, n in ev] + [(e, -1) for s, e, q, n in ev])
hist = collections.Counter(); cur = 0; last = 0.0; bubbles = []
for t, d in pts:
    hist[cur] += t - last
    if cur == 0 and t - last >= 5: bubbles.append((round(last, 1), round(t - last, 1)))
    cur += d; last = t
print(f"forward {end:.0f} us, {len(ev)} kernels on {len(set(q for s, e, q, n in ev))} queues; time with k kernels running:", {k: round(v) for k, v in sorted(hist.items())})
print("bubbles >= 5 us (start, length):", bubbles, "sum", round(sum(b for a, b in bubbles)))
