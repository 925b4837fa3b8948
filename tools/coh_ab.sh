#!/bin/bash
# developer script (GPU box): duration of sort_repair_kernel inside the replayed bench iteration for library variants
# (build/exp/lib<name>.so; "default" = the in-tree library).  usage: tools/coh_ab.sh <outdir> variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset FNX_RASTER_LIB; else export FNX_RASTER_LIB=$R/build/exp/lib$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$v -o r -- python $R/bench.py --no-cpu-baseline --steps 50 > $O/$v.json 2> $O/$v.err
  python $R/tools/iter_timeline.py $(find $O/tr_$v -name "*kernel_trace.csv" | head -1) > $O/$v.txt 2>&1
  rm -rf $O/tr_$v
  echo "== $v: $(grep -h 'window\|sort_repair\|preprocess' $O/$v.txt | tr '\n' ' ')"
done
