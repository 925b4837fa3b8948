#!/bin/bash
# developer script (GPU box).  usage: tools/trace_kernels.sh <name> [ENV=VAL ...] -- kernel durations (avg us) of one bench run under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
name=$1; shift
for e in "$@"; do export "$e"; done
rm -rf /tmp/tr_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$name -o x -- python /root/repo/bench.py --no-cpu-baseline --no-exact-leg --no-drop-in --steps 100 > /dev/null 2>&1
f=$(find /tmp/tr_$name -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print("$name", r["Name"][:60].ljust(60), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:8.1f}')
PY
