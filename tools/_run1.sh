export TMPDIR=/tmp; R=$PWD; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vprof -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline >/dev/null 2>&1; cd $R
python tools/iter_timeline.py gpurun_out/vprof/x_kernel_trace.csv full
