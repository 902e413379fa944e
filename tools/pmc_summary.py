"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel -> text table."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-70:]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(name, r["Counter_Name"])] += 1
for name, d in agg.items():
    if "cocos" not in name: continue
    print(name)
    for c, v in sorted(d.items()):
        n = cnt[(name, c)]
        print(f"    {c:32s} per-dispatch {v / n:16.1f}   (dispatches {n})")
