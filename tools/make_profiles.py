"""Build the committed summaries under profiles/ from the raw rocprofv3 output of tools/collect_profiles.sh
(gpurun_out/<tag>/).  Usage: python tools/make_profiles.py [tag, default r02; e.g. r02_config2] [output directory,
default profiles/].  collect_profiles.sh runs it on the GPU box (the raw traces are too large to travel back)."""
import collections, csv, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ARGS = open(os.path.join(ROOT, "gpurun_out", TAG, "args.txt")).read().strip() if os.path.exists(os.path.join(ROOT, "gpurun_out", TAG, "args.txt")) else ""
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)
sys.path.insert(0, ROOT)
from fluidnexus_amd.build import csrc_hash  # noqa: E402
CSRC_SHA = csrc_hash()


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    return n.split("(")[0]


# 1. bench line
line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(DST, f"{TAG}_bench.json"), "w"), indent=1)

# 2. kernel stats of the default bench
rows = list(csv.DictReader(open(os.path.join(SRC, "stats", "r_kernel_stats.csv"))))
with open(os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
with open(os.path.join(DST, f"{TAG}_bench_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py {ARGS} --no-cpu-baseline ({TAG})\n\n")
    f.write("The bench run: eager warm-up iterations, hipGraph capture, graph replays (timed region), then eager "
            "iterations with the in-library event hooks, one single-view render per view, the rasteriser-only timing and "
            "the KNN_K report (torch kernels); all views of an "
            f"iteration go through ONE launch of each rasteriser kernel.  Total kernel time in the trace {tot:.1f} ms.\n\n")
    f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:45]:
        f.write(f"| `{short(r['Name'])[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                f"{float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.2f} |\n")

# 3. timeline of one replayed iteration
tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "iter_timeline.py"),
                     os.path.join(SRC, "stats", "r_kernel_trace.csv"), "full"], capture_output=True, text=True).stdout
open(os.path.join(DST, f"{TAG}_iteration_timeline.txt"), "w").write(
    "# one replayed hipGraph iteration (rocprofv3 --kernel-trace of the default bench): per-kernel totals, then the\n"
    "# timeline: start us, gap to the previous kernel, duration, queue, kernel\n" + tl)


def pmc(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(SRC, dirname, "r_counter_collection.csv"))):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def durations(dirname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(SRC, dirname, "r_kernel_trace.csv"))):
        d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d


# 4. HBM traffic
fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) over `python bench.py "
                    + ARGS + " --steps 5 --warmup 2 --no-cpu-baseline --no-graph`; per-launch averages over ALL launches of a kernel in that "
                    "run (the rasteriser kernels also run single-view at the end of the bench; the blend backward only runs "
                    "view-batched, 5 views per launch). FETCH_SIZE is reported in KB and counts 64 B per 128 B request on gfx950 "
                    "(MI355X_MICROARCH.md, HBM section): fetch_bytes = 2 * FETCH_SIZE * 1024; write_bytes = WRITE_SIZE * 1024 "
                    "(uncalibrated; device-scope atomics are not counted)."}
for k in fetch:
    if ("fnx::" in k or "kernel" in k) and "at::" not in k:
        traffic[k] = {"fetch_bytes": 2 * fetch[k].get("FETCH_SIZE", 0.0) * 1024,
                      "write_bytes": write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024}
traffic["_csrc_sha16"] = CSRC_SHA  # the kernels these counters were collected on (bench.py: traffic_stale)
json.dump(traffic, open(os.path.join(DST, f"{TAG}_pmc_traffic.json"), "w"), indent=1)

# 5. SQ counters
sq, dur = pmc("pmc_sq"), durations("pmc_sq")
with open(os.path.join(DST, f"{TAG}_sq_counters.md"), "w") as f:
    f.write(f"# SQ counters per launch ({TAG}): rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES "
            f"SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -- python bench.py {ARGS} --no-cpu-baseline --no-graph "
            "--steps 5 --warmup 2\n\n"
            "`valu %` = SQ_INSTS_VALU / (duration x 933 G wave-instructions/s): share of the chip's MEASURED fp32 VALU issue rate "
            "(tools/micro/pk_rate.hip: dependent-free v_fma_f32 streams, 8 waves per SIMD, reach 933 G wave64 instructions/s = "
            "one per 2.63 cycles per SIMD at 2.4 GHz; v_pk_fma_f32 reaches 433 G/s, i.e. no more fp32 work per second); "
            "`busy %` = SQ_ACTIVE_INST_VALU x 4 / (duration x 2.4 GHz x 1024 SIMDs); `wait %` = SQ_WAIT_ANY / SQ_WAVE_CYCLES "
            "(waves parked in s_waitcnt / barriers); `stall %` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.  Durations are from the counter "
            "run (slower than the plain trace); averages over all launches (rasteriser forward kernels include single-view launches).\n\n")
    f.write("| kernel | us | VALU M | SALU M | LDS M | waves | valu % | busy % | stall % | wait % |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for k, c in sorted(sq.items(), key=lambda kv: -sum(dur[kv[0]]) / max(len(dur[kv[0]]), 1)):
        us = sum(dur[k]) / len(dur[k])
        if us < 8 or "at::" in k or "rocclr" in k:
            continue
        cap = us * 2400 / 4 * 1024
        wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        f.write(f"| `{k[:60]}` | {us:.1f} | {c.get('SQ_INSTS_VALU', 0) / 1e6:.2f} | {c.get('SQ_INSTS_SALU', 0) / 1e6:.2f} | "
                f"{c.get('SQ_INSTS_LDS', 0) / 1e6:.2f} | {c.get('SQ_WAVES', 0):.0f} | "
                f"{100 * c.get('SQ_INSTS_VALU', 0) / (us * 1e-6 * 933e9):.0f} | {100 * c.get('SQ_ACTIVE_INST_VALU', 0) / cap:.0f} | "
                f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} |\n")
# the same counters for bench.py (roofline.valu): per kernel the per-launch averages and the counter run's duration
def _derived(k, c):
    us = sum(dur[k]) / len(dur[k])
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    return dict(c, us=us, valu_busy=c.get("SQ_ACTIVE_INST_VALU", 0) / (us * 2400 / 4 * 1024),
                wait_frac=c.get("SQ_WAIT_ANY", 0) / wc, stall_frac=c.get("SQ_WAIT_INST_ANY", 0) / wc)


sq_json = {"_note": "per-launch averages of the SQ counter pass (see the .md of the same name); valu_busy = SQ_ACTIVE_INST_VALU x 4 / "
                    "(us x 2.4 GHz x 1024 SIMDs), wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; lds_busy (from the LDS pass) = "
                    "SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES",
           **{k: _derived(k, c) for k, c in sq.items() if "fnx::" in k or ("kernel" in k and "at::" not in k)}}
sq_json["_csrc_sha16"] = CSRC_SHA
json.dump(sq_json, open(os.path.join(DST, f"{TAG}_sq_counters.json"), "w"), indent=1)
# 6. LDS pipeline
if os.path.exists(os.path.join(SRC, "pmc_lds", "r_counter_collection.csv")):
    lds, dur = pmc("pmc_lds"), durations("pmc_lds")
    with open(os.path.join(DST, f"{TAG}_lds_counters.md"), "w") as f:
        f.write(f"# LDS pipeline per launch ({TAG}): rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES "
                f"SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -- python bench.py {ARGS} --no-cpu-baseline --no-graph --steps 5 "
                "--warmup 2\n\n`lds busy %` = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (cycles the compute units' LDS pipelines "
                "work, of the cycles the units are busy); `conflict %` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                "`cycles / instr` = SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS.\n\n"
                "| kernel | us | LDS instr M | lds busy % | conflict % | cycles / instr |\n|---|---|---|---|---|---|\n")
        for k, c in sorted(lds.items(), key=lambda kv: -sum(dur[kv[0]]) / max(len(dur[kv[0]]), 1)):
            us = sum(dur[k]) / len(dur[k])
            if us < 8 or "at::" in k or "rocclr" in k or not c.get("SQ_INSTS_LDS"):
                continue
            act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
            f.write(f"| `{k[:60]}` | {us:.1f} | {c['SQ_INSTS_LDS'] / 1e6:.2f} | {100 * act / max(c.get('SQ_BUSY_CU_CYCLES', 1), 1):.0f} | "
                    f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(act, 1):.0f} | {act / c['SQ_INSTS_LDS']:.1f} |\n")
    for k, c in lds.items():
        if k in sq_json and c.get("SQ_BUSY_CU_CYCLES"):
            sq_json[k]["lds_busy"] = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / c["SQ_BUSY_CU_CYCLES"]
    sq_json["_csrc_sha16"] = CSRC_SHA
    json.dump(sq_json, open(os.path.join(DST, f"{TAG}_sq_counters.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(DST)))
