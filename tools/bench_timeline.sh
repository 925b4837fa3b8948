#!/bin/bash
# developer script (GPU box): kernel timeline of one replayed bench iteration -> gpurun_out/<tag>/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o r -- python $R/bench.py --no-cpu-baseline --steps 50 "$@" > $O/bench.json 2> $O/bench.err
python $R/tools/iter_timeline.py $(find $O/tr -name "*kernel_trace.csv" | head -1) full spacings > $O/timeline.txt 2>&1
rm -rf $O/tr
python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'])"
cat $O/timeline.txt
