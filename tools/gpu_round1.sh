#!/bin/bash
# developer script: first GPU pass of the fast blend arithmetic
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fast_math_gpu.py -x -q -s -m gpu > gpurun_out/r3a/fast_tests.log 2>&1
echo "fast tests rc=$?" >> gpurun_out/r3a/fast_tests.log
timeout 900 python -m pytest tests/test_raster_parity_gpu.py tests/test_views_gpu.py -x -q -m gpu > gpurun_out/r3a/parity_tests.log 2>&1
echo "parity tests rc=$?" >> gpurun_out/r3a/parity_tests.log
for m in exact fast; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3a/lab_$m -o lab -- python $GRAFT_REPO_ROOT/tools/kernel_lab.py --iters 10 --math $m > $GRAFT_REPO_ROOT/gpurun_out/r3a/lab_$m.log 2>&1)
  f=$(find gpurun_out/r3a/lab_$m -name "*kernel_stats.csv" | head -1)
  python tools/kstats.py $f 16 >> gpurun_out/r3a/lab_$m.log 2>&1
done
for m in exact fast; do
  timeout 600 python bench.py --steps 100 --warmup 5 --blend-math $m --no-cpu-baseline > gpurun_out/r3a/bench_$m.json 2> gpurun_out/r3a/bench_$m.err
done
