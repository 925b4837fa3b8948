#!/bin/bash
# Developer tool: repeat a pytest selection N times and report the exit codes (hunting an intermittent abort).
# FNX_COLD=1 clears MIOpen's user caches before every run (a fresh GPU box starts with none).
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
N=$1; shift
mkdir -p $O
for i in $(seq 1 $N); do
  if [ "${FNX_COLD:-0}" = "1" ]; then rm -rf ~/.cache/miopen ~/.config/miopen; fi
  python -X faulthandler -m pytest "$@" -x -q -m gpu > $O/run_$i.log 2>&1
  echo "run $i rc=$?"
done
grep -l "Aborted\|Fatal" $O/run_*.log
ls -d ~/.cache/miopen ~/.config/miopen 2>&1 | head
