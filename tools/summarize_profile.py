#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the files kept under
profiles/: <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <name>_pmc_summary.json (counter means per
render_kernel launch + derived figures), <name>_bench.json and <name>_bench_under_rocprof.json.
The workload is whatever the bench runs of that tag rendered (their JSON lines name it).
Usage: tools/summarize_profile.py <tag> <name>     e.g.  tools/summarize_profile.py r03_chess r03_chess"""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import source_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
assert stats, "no kernel stats under " + src
shutil.copy(stats[0], os.path.join(dst, name + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, name + "_bench.json"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, name + "_bench_under_rocprof.json"))

# Each counter pass runs `bench.py --steps 2 --warmup 1`: five launches that learn the tile hand-out, the warm-up,
# then the two timed ones.  Only those two count (the learning launches are slower by design).
TIMED = 2
# the trace run is `bench.py --steps 5 --warmup 1`: the last five launches are the timed ones
trace = glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv"))[0]
shutil.copy(trace, os.path.join(dst, name + "_kernel_trace.csv"))
launch_ms = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 4)
             for r in csv.DictReader(open(trace)) if "render_kernel" in r["Kernel_Name"]]
under_trace = json.load(open(os.path.join(src, "bench_under_trace.json")))

tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
    timed = sorted({int(r["Dispatch_Id"]) for r in rows})[-TIMED:]
    for r in rows:
        if int(r["Dispatch_Id"]) not in timed:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
c = {k: round(tot[k] / cnt[k], 1) for k in sorted(tot)}
launches = max(cnt.values())
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
# executed f64 arithmetic: wave-instructions by class, an FMA counts two flops, times the mean number of lanes that
# were active when a VALU instruction issued (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU)
f64_insts = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
f64_flop_insts = f64_insts + c["SQ_INSTS_VALU_FMA_F64"]
lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
kernel_ms = sum(launch_ms[-5:]) / 5
derived = {
    "gpu_cycles_per_launch": round(cycles),
    "f64_valu_instructions_per_launch": round(f64_insts),
    "f64_share_of_valu_instructions": round(f64_insts / c["SQ_INSTS_VALU"], 3),
    "mean_active_lanes_per_valu_instruction": round(lanes, 2),
    "f64_flops_per_launch": round(f64_flop_insts * lanes),
    "f64_flops_note": "(ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64 wave-instructions) x mean active lanes; compares, selects, moves, conversions and min/max are not counted",
    "f64_tflops_at_trace_kernel_time": round(f64_flop_insts * lanes / (kernel_ms * 1e-3) / 1e12, 3),
    "valu_busy_fraction": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles), 3),
    "valu_busy_fraction_note": "SQ_ACTIVE_INST_VALU (quad-cycles in which a wave executes a VALU instruction) x 4 / (1024 SIMDs x GPU cycles); "
                               "no cost per instruction class is assumed. Per VALU instruction that is %.2f cycles on average "
                               "(f64 arithmetic issues over 4 cycles for 64 lanes, 32-bit operations over 2 when a second wave can take the other slot)" % (c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"]),
    "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 2),
    "fetch_bytes_per_launch": round(c["FETCH_SIZE"] * 1024),
    "write_bytes_per_launch": round(c["WRITE_SIZE"] * 1024),
    "traffic_note": "FETCH_SIZE/WRITE_SIZE are in KiB per dispatch (checked against a 24.9 MB fill kernel: WRITE_SIZE 24300). They count L2<->fabric requests, Infinity Cache hits included. The reads here are 8 B/lane, a width the guide calls uncalibrated; no doubling applied.",
}
json.dump({
    "kernel_source_hash": source_hash(),
    "command": "rocprofv3 --pmc <one group per run> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline <workload> (tools/profile_round.sh)",
    "workload": under_trace["config"]["workload"],
    "workload_note": "1 MI355X, per render_kernel launch (mean of the %d timed launches of each pass)" % launches,
    "kernel_trace": {
        "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
        "render_kernel_ms_per_launch": launch_ms,
        "note": "launches 1-5 learn the tile hand-out (image order with tile timing, then up to four trials) and launch 6 is the warm-up; "
                "the kernel-stats average covers all of them, the bench's HIP events only the timed five",
        "mean_of_timed_launches_ms": round(sum(launch_ms[-5:]) / 5, 4),
        "hip_event_mean_ms_same_run": under_trace["roofline"]["kernel_ms"]},
    "counters": c, "derived": derived}, open(os.path.join(dst, name + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
