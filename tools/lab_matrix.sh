#!/bin/bash
# developer script (GPU box): kernel_lab timing for "variant|args" specs (variant = default or a build/exp name)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1)); v=${spec%%|*}; a=${spec#*|}
  if [ "$v" = default ]; then unset FNX_RASTER_LIB; else export FNX_RASTER_LIB=$R/build/exp/lib$v.so; fi
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$i -o r -- python $R/tools/kernel_lab.py --iters 10 $a > $O/st_$i.log 2>&1
  echo "== [$spec]: $(grep 'ms per batched' $O/st_$i.log)" >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$i -name "*kernel_stats.csv" | head -1) 4 | grep -E "blend" >> $O/summary.txt 2>&1
  rm -rf $O/st_$i
done
cat $O/summary.txt
