"""Developer tool: build a variant of libfnx_raster.so with extra -D flags into build/exp/lib<name>.so
(usage: python tools/build_variant.py name [--patch fluidnexus_amd/csrc/lab/x.patch] -DFNX_EXP_X=1 ...).  --patch applies
a lab patch (git diff against the repository root) to a temporary copy of csrc/ first."""
import os, shutil, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "fluidnexus_amd", "csrc")
if "--patch" in sys.argv:
    i = sys.argv.index("--patch")
    patch = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
    tmp = tempfile.mkdtemp(prefix="fnx_variant_")
    shutil.copytree(C, os.path.join(tmp, "fluidnexus_amd", "csrc"))
    shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
    subprocess.check_call(["patch", "-p1", "-i", patch], cwd=tmp)
    C = os.path.join(tmp, "fluidnexus_amd", "csrc")
E = os.path.join(R, "build", "exp")
os.makedirs(E, exist_ok=True)
out = os.path.join(E, f"lib{sys.argv[1]}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value", *sys.argv[2:], "-o", out] +
                      [os.path.join(C, f) for f in ("raster_forward.hip", "raster_binning.hip", "raster_backward.hip",
                                                    "raster_api.hip")])
print(out)
