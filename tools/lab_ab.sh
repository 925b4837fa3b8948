#!/bin/bash
# developer script (GPU box): time tools/kernel_lab.py for several library variants (build/exp/lib<name>.so, "default" =
# the in-tree library).  usage: tools/lab_ab.sh <outdir> "<lab args>" variant...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
LAB="$1"; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset FNX_RASTER_LIB; else export FNX_RASTER_LIB=$R/build/exp/lib$v.so; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v -o r -- python $R/tools/kernel_lab.py --iters 10 $LAB > $O/st_$v.log 2>&1
  echo "== $v: $(grep 'ms per batched' $O/st_$v.log)" >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$v -name "*kernel_stats.csv" | head -1) 4 >> $O/summary.txt 2>&1
  rm -rf $O/st_$v
done
cat $O/summary.txt
