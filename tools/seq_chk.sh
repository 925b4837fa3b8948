#!/bin/bash
# developer script (GPU box): per-frame set-up of the sequence leg for a few flag sets
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-seqchk}; shift
mkdir -p $O
run() { name=$1; shift
  python $R/bench.py --no-cpu-baseline --frames 3 --iters-per-frame 250 "$@" > $O/$name.json 2> $O/$name.err
  python -c "import json;d=json.load(open('$O/$name.json'));s=d['sequence'];print('$name', round(d['value'],1), 'seq', round(s['seq_iters_per_s'],1), 'boundary', round(s['frame_boundary_ms'],1), s['setup_ms_by_frame'], s['segments_ms_per_frame'])" 2>&1 | tail -1
}
run nolegs --no-drop-in --no-exact-leg
run nolegs_dk0 --no-drop-in --no-exact-leg --deep-kernel 0
run exact_only --no-drop-in
run dropin_only --no-exact-leg
run all
run nolegs2 --no-drop-in --no-exact-leg
