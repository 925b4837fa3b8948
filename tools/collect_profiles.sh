#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: collects everything profiles/ is built from into
# gpurun_out/<tag>/.  usage: tools/collect_profiles.sh <tag> [bench.py arguments, e.g. --config 2]
# rocprofv3 counter passes are separate runs with --kernel-trace only (no sys/hip/hsa tracing).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}; shift
ARGS="$*"
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "$ARGS" > $O/args.txt
python $R/bench.py $ARGS > $O/bench.json 2> $O/bench.err
# (the profiled runs leave out the record's side legs -- exact-mode re-capture, drop-in loop -- so that the traces hold
# the timed configuration's kernels only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py $ARGS --no-cpu-baseline --no-exact-leg --no-drop-in > $O/stats.log 2>&1
EAGER="python $R/bench.py $ARGS --no-cpu-baseline --no-exact-leg --no-drop-in --no-graph --steps 5 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r -- $EAGER > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r -- $EAGER > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $O/pmc_sq -o r -- $EAGER > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $O/pmc_lds -o r -- $EAGER > $O/pmc_lds.log 2>&1
# summaries next to the raw output (gpurun_out/<tag>/summary/); the raw traces stay on the box
mkdir -p $O/summary
python $R/tools/make_profiles.py $TAG $O/summary > $O/make_profiles.log 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
ls $O $O/summary
tail -c 300 $O/bench.json
