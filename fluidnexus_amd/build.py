"""Builds the gfx950 shared libraries in-tree with hipcc (no JIT cache, no torch headers).

`python -m fluidnexus_amd.build` or `fluidnexus_amd.build.build_all()`; called by
__graft_entry__.build().  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")

# library name -> sources.  -ffp-contract=off is part of the numerics contract.
LIBS = {
    "libfnx_raster.so": ["raster_forward.hip", "raster_binning.hip", "raster_backward.hip", "raster_api.hip"],
    "libfnx_physics.so": ["physics.hip"],
    "libfnx_losses.so": ["losses.hip"],
}
# -fno-slp-vectorize: the SLP vectoriser turns scalar fp32 arithmetic into v_pk_{mul,add,fma}_f32; on gfx950 the
# VALU-bound blend kernels measured 19 % slower with them (issue stalls + register shuffling), DESIGN.md 4.2.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function", "-Wno-unused-value"]


def csrc_hash() -> str:
    """16 hex digits over every kernel source and header (and the compiler flags): what a counter profile under profiles/
    was collected on.  bench.py compares it with the stamp tools/make_profiles.py leaves in the counter files and marks
    them stale when the kernels have changed since."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = []
    for d in (_CSRC, os.path.join(_CSRC, "lab"), os.path.join(_HERE, "..", "include")):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(out: str, srcs: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = list(srcs) + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    deps += [os.path.join(_CSRC, "lab", f) for f in os.listdir(os.path.join(_CSRC, "lab")) if f.endswith(".h")]
    deps.append(os.path.join(_HERE, "..", "include", "fnx_raster.h"))
    deps.append(os.path.join(_HERE, "..", "include", "fnx_physics.h"))
    deps.append(os.path.join(_HERE, "..", "include", "fnx_losses.h"))
    deps.append(os.path.abspath(__file__))  # the compiler flags live here
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_all(force: bool = False, verbose: bool = False) -> list[str]:
    built = []
    for name, files in LIBS.items():
        out = os.path.join(_HERE, name)
        srcs = [os.path.join(_CSRC, f) for f in files]
        if force or _stale(out, srcs):
            cmd = [hipcc()] + FLAGS + ["-o", out] + srcs
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        built.append(out)
    return built


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print(p)
