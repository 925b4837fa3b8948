#!/bin/bash
# developer script (GPU box): a rank's emulated share of config 4 (2 views: rank 0; 1 view: rank 1) per deep-kernel mode
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wsemu}; shift
mkdir -p $O
for r in 0 1; do
  for dk in 0 3 4 1; do
    python $R/bench.py --config 4 --emulate-world 4 --emulate-rank $r --shared-terms last-rank --freeze --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 --deep-kernel $dk "$@" > $O/r${r}_dk$dk.json 2> $O/r${r}_dk$dk.err
    python -c "import json;d=json.load(open('$O/r${r}_dk$dk.json'));print('rank $r deep_kernel $dk', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
  done
done
