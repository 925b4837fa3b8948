#!/bin/bash
# developer script (GPU box): kernel stats + SQ counters of tools/kernel_lab.py for the given blend math modes
# usage: tools/lab_profile.sh <outdir under gpurun_out> "<lab args>" mode...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
LAB="$1"; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$m -o r -- python $R/tools/kernel_lab.py --iters 10 --math $m $LAB > $O/st_$m.log 2>&1
  echo "== $m kernel stats" >> $O/summary.txt
  grep "ms per batched" $O/st_$m.log >> $O/summary.txt
  python $R/tools/kstats.py $(find $O/st_$m -name "*kernel_stats.csv" | head -1) 14 >> $O/summary.txt 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $O/sq_$m -o r -- python $R/tools/kernel_lab.py --iters 5 --math $m $LAB > $O/sq_$m.log 2>&1
  echo "== $m SQ counters" >> $O/summary.txt
  python $R/tools/pmc_summary.py $(dirname $(find $O/sq_$m -name "*counter_collection.csv" | head -1)) blend >> $O/summary.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d $O/sq2_$m -o r -- python $R/tools/kernel_lab.py --iters 5 --math $m $LAB > $O/sq2_$m.log 2>&1
  python $R/tools/pmc_summary.py $(dirname $(find $O/sq2_$m -name "*counter_collection.csv" | head -1)) blend >> $O/summary.txt 2>&1
  rm -rf $O/st_$m $O/sq_$m $O/sq2_$m
done
cat $O/summary.txt
