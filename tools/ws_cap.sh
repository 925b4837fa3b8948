#!/bin/bash
# developer script (GPU box): config 3 (5 views) with the staging-wave kernel taking only the N deepest tiles of a view
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wscap}; shift
mkdir -p $O
for cap in 0 8 16 32 64 128; do
    if [ $cap = 0 ]; then dk=0; else dk=3; fi
    FNX_WS_MAX=$cap python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 --deep-kernel $dk "$@" > $O/cap$cap.json 2> $O/cap$cap.err
    python -c "import json;d=json.load(open('$O/cap$cap.json'));print('cap $cap', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
done
