#!/bin/bash
# developer script (GPU box): bench it/s for several argument sets.  usage: tools/bench_args.sh <outdir> "args1" "args2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
i=0
for a in "$@"; do
  i=$((i+1))
  python $R/bench.py --no-cpu-baseline --steps 200 $a > $O/b$i.json 2> $O/b$i.err
  python -c "import json;d=json.load(open('$O/b$i.json'));print('[$a]', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
done
