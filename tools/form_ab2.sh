#!/bin/bash
# developer script (GPU box): lab timing of library variants with a given FNX_BWD_FORM.  usage: tools/form_ab2.sh <outdir> "<lab args>" variant:form ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
LAB="$1"; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; f=${spec#*:}
  export FNX_BWD_FORM=$f
  if [ "$v" = default ]; then unset FNX_RASTER_LIB; else export FNX_RASTER_LIB=$R/build/exp/lib$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v$f -o r -- python $R/tools/kernel_lab.py --iters 10 $LAB > $O/st_$v$f.log 2>&1
  echo "== $v form $f: $(grep 'ms per batched' $O/st_$v$f.log)" >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$v$f -name "*kernel_stats.csv" | head -1) 3 >> $O/summary.txt 2>&1
  rm -rf $O/st_$v$f
done
cat $O/summary.txt
