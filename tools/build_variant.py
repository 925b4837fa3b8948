"""Developer tool: build a variant of libfnx_raster.so with extra -D flags into build/exp/lib<name>.so
(usage: python tools/build_variant.py name -DFNX_EXP_X=1 ...)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "fluidnexus_amd", "csrc")
E = os.path.join(R, "build", "exp")
os.makedirs(E, exist_ok=True)
out = os.path.join(E, f"lib{sys.argv[1]}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value", *sys.argv[2:], "-o", out] +
                      [os.path.join(C, f) for f in ("raster_forward.hip", "raster_binning.hip", "raster_backward.hip",
                                                    "raster_api.hip")])
print(out)
