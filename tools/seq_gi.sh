#!/bin/bash
# developer script (GPU box): the sequence leg's frame boundary by iterations per re-captured graph
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-seqgi}; shift
mkdir -p $O
for gi in 3 1 2 3 1; do
    python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 6 --iters-per-frame 250 --seq-graph-iters $gi "$@" > $O/gi$gi.json 2> $O/gi$gi.err
    python -c "import json;d=json.load(open('$O/gi$gi.json'));s=d['sequence'];print('graph iters $gi', round(d['value'],1), 'seq', round(s['seq_iters_per_s'],1), 'boundary ms', round(s['frame_boundary_ms'],2), 'vs', round(s['vs_steady_state'],4), s['segments_ms_per_frame'], s['setup_ms_by_frame'])" 2>&1 | tail -1
done
