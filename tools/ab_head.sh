# A/B on the GPU box: the working tree's bench against build/headtree's (a checkout of HEAD with its own libraries)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for w in headtree work; do
    if [ $w = work ]; then D=$R; else D=$R/build/headtree; fi
    ( cd $D && python bench.py --no-cpu-baseline --no-drop-in --frames 0 --no-exact-leg --repeats 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],1), [round(x,1) for x in d['spread']['iters_per_s']], {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items()})" )
  done
done
