#!/bin/bash
# developer script (GPU box): sequence leg (--frames 3 --iters-per-frame N) for "name:extra bench args" variants.
# usage: tools/seq_ab2.sh <outdir> <iters-per-frame> "name:args" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
N=$1; shift
mkdir -p $O
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  python $R/bench.py --frames 3 --iters-per-frame $N --no-cpu-baseline --no-drop-in --no-exact-leg --steps 100 $args > $O/$name.json 2> $O/$name.err
  python -c "import json;d=json.load(open('$O/$name.json'));s=d['sequence'];print('$name', 'steady', round(d['value'],1), 'seq', round(s['seq_iters_per_s'],1), 'boundary ms', round(s['frame_boundary_ms'],2), 'vs', round(s['vs_steady_state'],4), {k: round(v,2) for k,v in s['segments_ms_per_frame'].items()})" 2>&1 | tail -1
done
