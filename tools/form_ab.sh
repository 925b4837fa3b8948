#!/bin/bash
# developer script (GPU box): the two forms of the blend backward (FNX_BWD_FORM=0 rows / 1 lanes) on the lab scene and in
# the bench.  usage: tools/form_ab.sh <outdir> [lab args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
LAB="${1:---mode3 --math fast}"
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  export FNX_BWD_FORM=$f
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$f -o r -- python $R/tools/kernel_lab.py --iters 10 $LAB > $O/st_$f.log 2>&1
  echo "== form $f: $(grep 'ms per batched' $O/st_$f.log)" >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$f -name "*kernel_stats.csv" | head -1) 4 >> $O/summary.txt 2>&1
  rm -rf $O/st_$f
done
cat $O/summary.txt
