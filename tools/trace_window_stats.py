"""Per-kernel totals of a rocprofv3 kernel trace restricted to the LAST fraction of the traced time (steady state:
MIOpen's find phase and warm-up runs of solvers sit in the first part).  Usage: trace_window_stats.py trace.csv [frac]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    if int(r["Start_Timestamp"]) < cut:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r["Kernel_Name"].replace("void ", "")[:100]]
    a[0] += 1
    a[1] += d
    tot += d
print(f"# window = last {frac:.0%} of the trace = {(t1 - cut) / 1e6:.1f} ms wall, kernel time {tot / 1e3:.1f} ms")
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:6d} {us:12.1f} us {us / tot * 100:6.2f} %  {name}")
