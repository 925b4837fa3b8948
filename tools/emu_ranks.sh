#!/bin/bash
# developer script (GPU box): every rank's share of a view-sharded run on one GPU, no communication
# (bench.py --emulate-world W --emulate-rank r), for the ways of placing the view-independent terms.
# usage: tools/emu_ranks.sh <outdir> <config> <world> mode...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
CFG=$1; shift
W=$1; shift
mkdir -p $O
for mode in "$@"; do
  line="config $CFG world $W $mode:"
  for r in $(seq 0 $((W-1))); do
    python $R/bench.py --config $CFG --emulate-world $W --emulate-rank $r --shared-terms $mode --freeze --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 > $O/c${CFG}_w${W}_${mode}_r$r.json 2> $O/c${CFG}_w${W}_${mode}_r$r.err
    v=$(python -c "import json;print(round(json.load(open('$O/c${CFG}_w${W}_${mode}_r$r.json'))['value'],1))" 2>/dev/null)
    line="$line r$r=$v"
  done
  echo "$line"
done
