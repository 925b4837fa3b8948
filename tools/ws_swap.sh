#!/bin/bash
# developer script (GPU box): staging-wave kernel for the deep tiles with the kernels' streams swapped, by cap and view count
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wsswap}; shift
mkdir -p $O
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python $R/bench.py --no-cpu-baseline --no-drop-in --no-exact-leg --frames 0 --steps 200 "$@" > $O/$name.json 2> $O/$name.err
  python -c "import json;d=json.load(open('$O/$name.json'));print('$name', round(d['value'],1), 'it/s', {k: round(v,1) for k,v in d['roofline']['other_kernels_avg_us'].items() if v})" 2>&1 | tail -1
}
run v5_base X=1 -- --deep-kernel 0
for cap in 16 64 128 512; do run v5_swap_cap$cap FNX_WS_MAX=$cap -- --deep-kernel 3; done
run v5_noswap_cap64 FNX_WS_MAX=64 FNX_WS_SWAP=0 -- --deep-kernel 3
E2="--config 4 --emulate-world 2 --emulate-rank 0 --shared-terms last-rank --freeze"
run v3_base X=1 -- $E2 --deep-kernel 0
run v3_swap X=1 -- $E2 --deep-kernel 3
run v3_swap_cap64 FNX_WS_MAX=64 -- $E2 --deep-kernel 3
run v3_noswap FNX_WS_SWAP=0 -- $E2 --deep-kernel 3
E4="--config 4 --emulate-world 4 --emulate-rank 0 --shared-terms last-rank --freeze"
run v2_all X=1 -- $E4 --deep-kernel 4
run v2_swap X=1 -- $E4 --deep-kernel 3
