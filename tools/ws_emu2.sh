#!/bin/bash
# developer script (GPU box): rank 0's emulated share of config 3's five views on 2 ranks (3 views) per deep-kernel mode
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wsemu2}; shift
mkdir -p $O
for dk in 0 3 4 0 4; do
    python $R/bench.py --config 4 --emulate-world 2 --emulate-rank 0 --shared-terms last-rank --freeze --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 --deep-kernel $dk "$@" > $O/dk$dk.json 2> $O/dk$dk.err
    python -c "import json;d=json.load(open('$O/dk$dk.json'));print('3 views deep_kernel $dk', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
done
