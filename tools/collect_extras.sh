#!/bin/bash
# Run on the GPU box: side records of the round (one bench line each) -> gpurun_out/<tag>_extras/<tag>_extra_<name>.json
# usage: tools/collect_extras.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
O=$R/gpurun_out/${TAG}_extras
mkdir -p $O
run() {  # name, env assignments (comma separated, may be empty), bench arguments
  local name=$1 envs=$2; shift 2
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done
    timeout 300 python $R/bench.py --no-cpu-baseline --no-drop-in --frames 0 "$@" > $O/${TAG}_extra_$name.json 2> $O/$name.err )
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_extra_$name.json"))
    print("$name", round(d["value"], 1), d["unit"], round(d["ms_per_step"] * 1e3, 1), "us")
except Exception as e:
    print("$name FAILED", e)
PY
}
run c3_1000 "" --steps 1000
run c3_exact "" --blend-math exact
run c3_eager "" --no-graph
run c3_first "" --stage first
run c3_nodist "" --no-distance
run c3_nosplit "" --no-static-split
run c3_r01scene "" --scene r01
run c3_screen_grad "FNX_SCREEN_GRAD=1"
run c3_dist1 "FNX_FORCE_DIST=1"
run c3_dist1_eager "FNX_FORCE_DIST=1,FNX_GRAPH_ALLREDUCE=0"
run c3_verlet "FNX_DIST_VERLET=1"
run c3_no_knn_watch "FNX_KNN_WATCH=0"
run c3_graph5 "" --graph-iters 5
run c3_capture_rebuilds_grid "FNX_CAPTURE_KEEP_GRID=0"
run c5_two_renders "FNX_DUAL_FUSED=0" --config 5
run c5_coherent_always "" --config 5 --sort coherent-always
run c5_loss_side_stream "FNX_DUAL_LOSS_STREAM=1" --config 5
run c3_radix "" --sort radix
run c3_dist_grid "FNX_DIST_GRID=1"
run c3_seq250 "" --frames 3 --iters-per-frame 250 --no-exact-leg
run c3_seq1000 "" --frames 3 --iters-per-frame 1000 --no-exact-leg
run c3_full_geometry "" --full-geometry --sort-four-passes
run c4_emu4 "" --config 4 --emulate-world 4
run c4_emu4_deep "" --config 4 --emulate-world 4 --deep-kernel 1
run c5_emu8 "" --config 5 --emulate-world 8
run c3_emu5 "" --config 3 --emulate-world 5
run c3_emu5_deep "" --config 3 --emulate-world 5 --deep-kernel 2
run c2_exact "" --config 2 --blend-math exact
run visual_exact "" --stage visual --blend-math exact
rm -f $O/*.err
ls $O
