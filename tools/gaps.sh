#!/bin/bash
# developer script (GPU box): the gaps on the main queue of one replayed iteration.  usage: tools/gaps.sh <name> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
name=$1; shift
for e in "$@"; do export "$e"; done
rm -rf /tmp/gp_$name
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$name -o x -- python /root/repo/bench.py --no-cpu-baseline --no-exact-leg --no-drop-in --steps 100 > /dev/null 2>&1
f=$(find /tmp/gp_$name -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the replayed part: the last 40 % of the trace; main queue = the queue of blend_forward
tail = rows[int(len(rows) * 0.5):]
q = collections.Counter(r["Queue_Id"] for r in tail if "blend_forward" in r["Kernel_Name"]).most_common(1)[0][0]
main = [r for r in tail if r["Queue_Id"] == q]
gaps = collections.defaultdict(list)
for a, b in zip(main, main[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[(a["Kernel_Name"][:28], b["Kernel_Name"][:28])].append(g)
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    m = sorted(v)[len(v) // 2]
    if len(v) >= 10 and m > 0.5:
        print("$name", f"{m:6.1f} us median gap  x{len(v):4d}  {k[0]} -> {k[1]}")
        tot += m
starts = [int(r["Start_Timestamp"]) for r in main if "adam_count_scan" in r["Kernel_Name"]]
per = sorted((b - a) / 1e3 for a, b in zip(starts, starts[1:]))
if per:
    print("$name iteration period (adam -> adam), median / quartiles:", round(per[len(per) // 2], 1), round(per[len(per) // 4], 1), round(per[3 * len(per) // 4], 1), "us over", len(per))
dur = collections.defaultdict(list)
for r in main:
    dur[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$name main-queue kernels (median us):", ", ".join(f"{k.split('::')[-1][:22]} {sorted(v)[len(v)//2]:.1f}" for k, v in dur.items()))
print("$name total of the median gaps > 0.5 us:", round(tot, 1), "| rows", len(rows), "main", len(main), "queue", q, "pairs", len(gaps))
PY
