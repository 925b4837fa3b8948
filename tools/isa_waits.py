"""Developer tool: the global-memory loads and the waits for them of one kernel, in program order (hipcc -S), to find
loads that are waited for one at a time -- `if (i < n) x = p[i]` in an unrolled loop gives every load a branch of its own
with s_waitcnt vmcnt(0) behind it.  usage: python tools/isa_waits.py raster_binning.hip emit_kernel"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "fluidnexus_amd", "csrc", sys.argv[1])
out = "/tmp/isa_" + os.path.basename(src) + ".s"
if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fno-slp-vectorize", "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)
txt = open(out).read().splitlines()
pat = sys.argv[2]
i = 0
while i < len(txt):
    l = txt[i]
    if l.startswith("_Z") and ": ;" in l:
        name = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout.strip()
        if pat in name:
            j = next(k for k in range(i, len(txt)) if txt[k].startswith(".Lfunc_end"))
            print("==", name[:120], j - i, "lines")
            for k in range(i, j):
                s = txt[k].strip()
                if any(x in s for x in ("global_load", "global_store", "global_atomic", "s_waitcnt vmcnt", "s_barrier", "buffer_", "s_load_dword")) or re.match(r"\.LBB\d+_\d+:.*Loop", s):
                    print(f"{k - i:6d}  {s[:100]}")
            i = j
    i += 1
