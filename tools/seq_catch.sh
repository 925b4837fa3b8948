#!/bin/bash
# developer script (GPU box): several sequence runs, set-up split of the slowest frames
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-seqcatch}; shift
mkdir -p $O
for i in 1 2; do
  python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 60 --iters-per-frame 250 "$@" > $O/run$i.json 2> $O/run$i.err
  python - <<PY
import json
s=json.load(open("$O/run$i.json"))["sequence"]
su=s["setup_ms_by_frame"]; sp=s["setup_split_ms_by_frame"]
print("run $i", round(s["seq_iters_per_s"],1), "median setup", sorted(su)[len(su)//2], "frames > 25 ms:", sum(1 for x in su if x > 25))
slow=[k for k,x in enumerate(su) if x > 25][:4]
sg=s["allocator_segments_by_frame"]
for k in slow: print("   frame", k, su[k], sp[k], sg[k])
print("   typical", sp[5], sg[5], sg[30])
PY
done
