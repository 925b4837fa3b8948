import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
os.environ["FNX_RASTER_LIB"] = "/root/repo/build/exp/liblanes.so"
sys.argv = ["kernel_lab", "--no-backward", "--iters", "1"]
exec(open("/root/repo/tools/kernel_lab.py").read())
buf = (C.c_ulonglong * 8)()
lib.fnx_debug_lane_stats(buf)
ev, live, hit, nd, b44, anyl = [buf[i] for i in range(6)]
print("wave-evaluations", ev, "live lanes/eval", live / ev, "hit lanes/eval", hit / ev, "not-done lanes/eval", nd / ev,
      "4x4 blocks with live lane per eval", b44 / ev, "evals with any live lane", anyl / ev)
