"""Developer tool: per-kernel averages of a rocprofv3 --pmc counter_collection CSV (+ durations from the kernel trace)."""
import collections, csv, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(d + "/r_counter_collection.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(d + "/r_kernel_trace.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pat = sys.argv[2] if len(sys.argv) > 2 else "blend"
for n, c in agg.items():
    if pat not in n:
        continue
    us = sum(dur[n][-5:]) / max(len(dur[n][-5:]), 1)
    print(n[:60], "last-5 avg us %.1f" % us, {k: "%.3g" % (sum(v[-5:]) / len(v[-5:])) for k, v in c.items()})
