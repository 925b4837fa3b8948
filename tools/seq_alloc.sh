#!/bin/bash
# developer script (GPU box): the sequence leg with the caching allocator rounding sizes up (sizes grow by 0.3 % per frame)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-seqalloc}; shift
mkdir -p $O
for conf in "" "max_split_size_mb:32" "roundup_power2_divisions:8,max_split_size_mb:32" "roundup_power2_divisions:4,max_split_size_mb:20"; do
  name=$(echo "x$conf" | tr -c 'a-zA-Z0-9\n' '_')
  PYTORCH_HIP_ALLOC_CONF="$conf" PYTORCH_CUDA_ALLOC_CONF="$conf" python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 50 --iters-per-frame 120 "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); s=d["sequence"]
    su=s["setup_ms_by_frame"]; sg=s["allocator_segments_by_frame"]
    print("conf [$conf]", round(d["value"],1), "seq", round(s["seq_iters_per_s"],1), "vs", round(s["vs_steady_state"],3), "median setup", sorted(su)[len(su)//2], "max", max(su), "frames > 25 ms:", sum(1 for x in su if x > 25))
    print("   segments allocated per frame:", [g[0] for g in sg][:40], "reserved GiB first/last", sg[0][2], sg[-1][2])
except Exception as e:
    print("conf [$conf] FAILED", e); print(open("$O/$name.err").read()[-600:])
PY
done
