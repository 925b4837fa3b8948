#!/bin/bash
# developer script (GPU box): bench it/s for "name:ENV=VAL,ENV=VAL" variants.  usage: tools/bench_ab.sh <outdir> "<bench args>" spec...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
ARGS="$1"; shift
mkdir -p $O
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS; python $R/bench.py --no-cpu-baseline --steps 200 $ARGS > $O/$name.json 2> $O/$name.err )
  python -c "import json;d=json.load(open('$O/$name.json'));print('$name', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items()})" 2>&1 | tail -1
done
