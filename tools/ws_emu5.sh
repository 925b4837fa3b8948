#!/bin/bash
# developer script (GPU box): a rank's emulated share of config 5 on 8 ranks (one view, dual mode) per deep-kernel mode
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wsemu5}; shift
mkdir -p $O
for r in 0 3; do
  for dk in 0 5 3; do
    python $R/bench.py --config 5 --emulate-world 8 --emulate-rank $r --shared-terms last-rank --freeze --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 --deep-kernel $dk "$@" > $O/r${r}_dk$dk.json 2> $O/r${r}_dk$dk.err
    python -c "import json;d=json.load(open('$O/r${r}_dk$dk.json'));print('config 5 rank $r of 8 deep_kernel $dk', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
  done
done
python $R/bench.py --config 5 --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 100 > $O/c5.json 2> $O/c5.err
python -c "import json;d=json.load(open('$O/c5.json'));print('config 5 single GPU', round(d['value'],1))"
