# robustness of the in-graph all-reduce capture: N short forced-dist runs, count the failures
ok=0; bad=0
for i in $(seq 1 ${1:-20}); do FNX_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-drop-in --frames 0 --no-exact-leg --repeats 1 --steps 40 > gpurun_out/d1_run.json 2> gpurun_out/d1_run.err; rc=$?
  if [ $rc -eq 0 ] && ! grep -q "eager collective" gpurun_out/d1_run.json gpurun_out/d1_run.err; then ok=$((ok+1)); else bad=$((bad+1)); cp gpurun_out/d1_run.err gpurun_out/d1_bad_$i.err; cp gpurun_out/d1_run.json gpurun_out/d1_bad_$i.json; echo "run $i rc=$rc"; fi; done
echo "ok=$ok bad=$bad"
