"""Developer tool: top kernels of a rocprofv3 --stats CSV (name shortened, calls, average / total microseconds)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{float(r['AverageNs']) / 1e3:9.1f} us avg {int(r['Calls']):5d} calls {float(r['TotalDurationNs']) / 1e6:8.2f} ms  {name[:70]}")
