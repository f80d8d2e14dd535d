"""Per-queue picture of a split-mode kernel trace: tools/trace_parts.py <rocprofv3 output dir>
(first start, last end, kernels, busy time of every queue; how much of the wall clock had 1, 2, 3, 4 sampler kernels in flight)"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wn_dense_grad" in r["Kernel_Name"] or "wn_split_tick" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
q = collections.defaultdict(list)
for r in rows:
    q[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, "G" if "grad" in r["Kernel_Name"] else "T"))
for k, v in sorted(q.items()):
    g = [b - a for a, b, n in v if n == "G" and b - a > 8000]
    t = [b - a for a, b, n in v if n == "T" and b - a > 5000]
    print("queue %s: %d kernels, %.1f .. %.1f us, busy %.1f us; live GEMMs %d avg %.1f us, live ticks %d avg %.1f us" % (
        k, len(v), min(a for a, b, n in v) / 1e3, max(b for a, b, n in v) / 1e3, sum(b - a for a, b, n in v) / 1e3,
        len(g), sum(g) / max(1, len(g)) / 1e3, len(t), sum(t) / max(1, len(t)) / 1e3))
ev = sorted([(a, 1) for v in q.values() for a, b, n in v] + [(b, -1) for v in q.values() for a, b, n in v])
lvl = 0; last = 0; hist = collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print("kernels in flight:", "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
mid = sorted((a, b, n, k) for k, v in q.items() for a, b, n in v)
mid = mid[len(mid) // 2: len(mid) // 2 + 24]
for a, b, n, k in mid:
    print("  q%s %s %9.1f -> %9.1f (%.1f us)" % (k, n, a / 1e3, b / 1e3, (b - a) / 1e3))
