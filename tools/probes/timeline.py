"""Timeline statistics of one forward from a rocprofv3 kernel trace: time with 0 / 1 / 2 / 3 / 4 kernels running, the bubbles."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("stem_")]
i0, i1 = st[len(st) // 2], st[len(st) // 2 + 1]
fw = rows[i0:i1]
t0 = int(fw[0]["Start_Timestamp"])
ev = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Kernel_Name"]) for r in fw]
dec = next((s for s, e, q, n in ev if n.startswith(("stage_average", "peaks_region"))), ev[-1][1])
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

# the phases with ONE kernel running, by kernel (where the chip runs a single launch: what the lanes cannot hide)
one = collections.Counter(); cur = []; last = 0.0
for t, d, n in sorted([(s, 1, n) for s, e, q, n in ev] + [(e, -1, n) for s, e, q, n in ev]):
    if len(cur) == 1: one[cur[0].split("(")[0][:60]] += t - last
    if d > 0: cur.append(n)
    else: cur.remove(n)
    last = t
print("single-kernel time by kernel (us):", {k: round(v) for k, v in one.most_common(12)}, "sum", round(sum(one.values())))
# around each bubble: the launches that end last before it and start first after it (queue, name)
for b0, bl in bubbles:
    before = sorted([x for x in ev if x[1] <= b0 + 0.05], key=lambda x: -x[1])[:3]
    after = sorted([x for x in ev if x[0] >= b0 + bl - 0.05], key=lambda x: x[0])[:4]
    print(f"bubble at {b0} (+{bl}):  before:", [(round(e, 1), q, n.split('(')[0][-40:]) for s, e, q, n in before], " after:", [(round(s, 1), q, n.split('(')[0][-40:]) for s, e, q, n in after])
# per-lane Gantt of the forward (start - end, duration, gap to the lane's previous launch): python tools/probes/timeline.py trace.csv gantt
if len(sys.argv) > 2 and sys.argv[2] == "gantt":
    def short(n):
        return n.split("(")[0].replace("void ", "").replace("conv_mfma_kernel", "conv")[:34]
    for q in sorted(set(e[2] for e in ev)):
        print(f"--- queue {q}")
        prev = None
        for s, e, qq, n in ev:
            if qq != q: continue
            print(f"{s:8.1f} {e:8.1f} {e - s:6.1f} {('+%.0f' % (s - prev)) if prev is not None else '':>6s}  {short(n)}")
            prev = e
