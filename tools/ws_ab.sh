#!/bin/bash
# developer script (GPU box): config-3 bench with the staging-wave forward (deep_kernel 3 / 4) against the per-tile kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ws}; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dk in 0 3 4 0 3; do
  python $R/bench.py --no-cpu-baseline --no-exact-leg --no-drop-in --frames 0 --steps 200 --deep-kernel $dk "$@" > $O/dk$dk.json 2> $O/dk$dk.err
  python -c "import json;d=json.load(open('$O/dk$dk.json'));print('deep_kernel $dk', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items()})" 2>&1 | tail -1
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st3 -o r -- python $R/bench.py --no-cpu-baseline --no-exact-leg --no-drop-in --frames 0 --steps 100 --deep-kernel 3 "$@" > $O/st3.log 2>&1
python $R/tools/kstats.py $(find $O/st3 -name "*kernel_stats.csv" | head -1) 8
rm -rf $O/st3
