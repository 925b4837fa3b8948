# developer script (GPU box): the sequence leg's frame boundary for several set-up variants
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in ${SEQ_VARIANTS:-"2 1 5" "2 0 5" "2 0 2" "2 0 3" "2 0 4" "1 0 3"}; do
  set -- $(echo $v | tr "_" " ")
  python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --repeats 1 --frames 4 --iters-per-frame 250 --seq-eager $1 --seq-capture-warmup $2 --seq-graph-iters $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sequence']; print('eager $1 warmup $2 graph $3:', round(s['seq_iters_per_s'],1), round(s['vs_steady_state'],3), round(s['frame_boundary_ms'],1), {k: round(x,1) for k,x in s['segments_ms_per_frame'].items()}, 'radix frames', s['frames_on_radix_sort'])"
done
