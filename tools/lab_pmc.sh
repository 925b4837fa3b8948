#!/bin/bash
# developer script (GPU box): PMC counters of the blend kernels in tools/kernel_lab.py.  usage: tools/lab_pmc.sh <outdir> "<lab args>" COUNTER...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
LAB="$1"; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc -o r -- python $R/tools/kernel_lab.py --iters 3 $LAB > $O/pmc.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$O/pmc/**/r_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "blend" in n or "emit" in n or "visual" in n:
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:70], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
rm -rf $O/pmc
