"""Developer tool: per-kernel register / spill / LDS summary of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py raster_forward.hip [pattern] [extra flags]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "fluidnexus_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
       "-Wno-unused-value", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage", *sys.argv[3:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*? Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print(f"{k:70s} vgpr {v.get('VGPRs')} spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')} lds {v.get('LDS Size [bytes/block]')}")
