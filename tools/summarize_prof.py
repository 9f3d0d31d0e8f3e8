"""Summarise rocprofv3 csv outputs (kernel stats + PMC counters per kernel name)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))
for f in find("*kernel_stats.csv"):
    print("== kernel stats", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %6s total_ns %14s avg_ns %12s pct %s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for f in find("*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:40]
        agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0)
    print("== counters", os.path.relpath(f, out))
    for k, d in agg.items():
        print("  ", k)
        for c, v in sorted(d.items()):
            print("      %-28s %.6g" % (c, v))
