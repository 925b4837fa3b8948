#!/bin/bash
# developer script (GPU box): time tools/kernel_lab.py for several argument sets.  usage: tools/lab_args.sh <outdir> "args1" "args2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$i -o r -- python $R/tools/kernel_lab.py --iters 10 $a > $O/st_$i.log 2>&1
  echo "== [$a]: $(grep 'ms per batched' $O/st_$i.log)" >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$i -name "*kernel_stats.csv" | head -1) 5 >> $O/summary.txt 2>&1
  rm -rf $O/st_$i
done
cat $O/summary.txt
