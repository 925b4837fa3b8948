"""Developer tool: run a few eager iterations of a bench configuration with the coherent sort and print its counters (calls,
fallbacks, why-bits per view).  usage: python tools/sort_why.py [bench args]"""
import subprocess, sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-drop-in", "--no-exact-leg", "--no-graph", "--steps", "6", "--warmup", "6"] + sys.argv[1:]
import bench
from fluidnexus_amd import rasterizer
from fluidnexus_amd.renderer import pipes
real = rasterizer.coherent_sort_counters
rasterizer.coherent_sort_counters = lambda: (0, 0)  # keep the mode on whatever happens
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for vb, _, _ in pipes._VIEW_BATCH_CACHE.values():
    for key in list(vb._sort_state):
        print("sort state", key, vb.sort_counters(key[0], key[1], why=True, sort_key=key[2]))
