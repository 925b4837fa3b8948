#!/bin/bash
# developer script (GPU box): bench it/s of the in-tree raster library against build/exp/lib<name>.so variants, interleaved.
# usage: tools/lib_ab.sh <outdir> "<bench args>" variant... ("default" = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
ARGS="$1"; shift
mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset FNX_RASTER_LIB; else export FNX_RASTER_LIB=$R/build/exp/lib$v.so; fi
  python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 $ARGS > $O/$v$rep.json 2> $O/$v$rep.err
  python -c "import json;d=json.load(open('$O/$v$rep.json'));print('$v', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
done
done
