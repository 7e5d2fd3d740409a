#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one round (kernel trace + counter passes, merged back under
gpurun_out/) into the tracked summaries under profiles/:  <tag>_kernel_stats.csv,
<tag>_counters.csv, <tag>_summary.md and traffic.json (HBM-side bytes per launch of the dominant
kernel, read by bench.py).  Usage: python profiles/summarize.py r1 <kernel-name> [<regex for the instantiation>]"""
import csv
import glob
import json
import os
import shutil
import sys

import re

tag, kernel = sys.argv[1], sys.argv[2]
# which instantiation: by default the one bench.py's headline leg launches (no counters, not the exact-cull variant, whose
# template arguments end in "true"); a third argument overrides the pattern
pattern = re.compile(sys.argv[3] if len(sys.argv) > 3 else re.escape(kernel) + r"<\d+, ?false, ?\d+, ?\d+, ?(true|false)(, ?false)?>")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")

newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]  # gpurun merges results into what earlier calls left: only the last run of a pass counts
stats = newest(glob.glob(os.path.join(src, f"{tag}_trace", "*", "*_kernel_stats.csv")))
avg_ns, chosen = None, None
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    best_total = -1.0
    for row in csv.DictReader(open(stats[0])):
        if pattern.search(row["Name"]) and float(row["TotalDurationNs"]) > best_total:  # several instantiations may match: the one the time went to
            best_total, avg_ns, chosen = float(row["TotalDurationNs"]), float(row["AverageNs"]), row["Name"]

m_inst = re.search(re.escape(kernel) + r"<[^>]*>", chosen) if chosen else None
chosen_inst = m_inst.group(0) if m_inst else None  # e.g. "bvh4_closest_hit_kernel<3, false, 2, 1, true, false>"
vals = {}
for d in (f"{tag}_pmc", f"{tag}_pmc_mem"):
    passes = {}
    for f in glob.glob(os.path.join(src, d, "*", "*", "*_counter_collection.csv")):
        passes.setdefault(os.path.dirname(f), []).append(f)
    for f in [g for files in passes.values() for g in newest(files)]:
        for row in csv.DictReader(open(f)):
            if (chosen_inst in row["Kernel_Name"]) if chosen_inst else pattern.search(row["Kernel_Name"]):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
with open(os.path.join(dst, f"{tag}_counters.csv"), "w") as fh:
    fh.write("counter,mean_per_launch,launches\n")
    for k in sorted(mean):
        fh.write(f"{k},{mean[k]:.6g},{len(vals[k])}\n")

bench = {}
p = os.path.join(src, f"{tag}_bench.json")
if os.path.exists(p):
    bench = json.load(open(p))
    shutil.copy(p, os.path.join(dst, f"{tag}_bench.json"))

lines = [f"# {tag}: rocprofv3 summary for `{kernel}` (bench.py default command, one MI355X)", ""]
if avg_ns:
    lines.append(f"* kernel-trace average duration: **{avg_ns / 1e6:.3f} ms** per launch "
                 f"(library HIP events in the same run: {bench.get('roofline', {}).get('kernel_ms', float('nan')):.3f} ms)")
if mean:
    cyc = mean.get("GRBM_GUI_ACTIVE", 0) / 8.0
    fetch = mean.get("FETCH_SIZE", 0) * 1024 * 2  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (calibrated on a
    write = mean.get("WRITE_SIZE", 0) * 1024      # random 128-B row gather, profiles/fetch_calib.hip: factor 2.000)
    traffic = fetch + write
    ms = (avg_ns or 0) / 1e6
    sys.path.insert(0, root)
    from bench import source_hash  # the profile describes this build only: bench.py ignores it once the kernel sources change
    simd_cycles = 256 * 4 * cyc    # SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles), summed over the SIMDs
    json.dump({"source_hash": source_hash(), "profile": f"profiles/{tag}_counters.csv",
               "hbm_bytes_per_launch": traffic, "fetch_bytes": fetch, "write_bytes": write,
               "l2_hit_rate": mean.get("TCC_HIT_sum", 0) / max(1.0, mean.get("TCC_HIT_sum", 0) + mean.get("TCC_MISS_sum", 0)),
               "valu_busy_frac": 4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, simd_cycles),
               "valu_lane_utilisation": mean.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * mean.get("SQ_ACTIVE_INST_VALU", 0)),
               "wave_wait_frac": mean.get("SQ_WAIT_ANY", 0) / max(1.0, mean.get("SQ_WAVE_CYCLES", 0)),
               "source": f"rocprofv3 --pmc, separate passes, {tag}; FETCH_SIZE doubled per the gfx950 "
                         "calibration (profiles/fetch_calib.hip measured 1.9996)"}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    lines += [
        f"* shader clock during the kernel: {cyc / (ms * 1e-3) / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 / duration)" if ms else "",
        f"* fabric-side traffic (L2 <-> Infinity Cache / HBM): {traffic / 1e9:.1f} GB per launch = "
        f"**{traffic / (ms * 1e-3) / 1e12:.2f} TB/s**" if ms else "",
        f"* L2 hit rate: {100 * mean.get('TCC_HIT_sum', 0) / max(1.0, mean.get('TCC_HIT_sum', 0) + mean.get('TCC_MISS_sum', 0)):.1f} % "
        f"({mean.get('TCC_REQ_sum', 0):.3g} requests)",
        f"* texture addresser busy: {100 * mean.get('TA_TA_BUSY_sum', 0) / max(1.0, 256 * cyc):.0f} % of 256 CUs x kernel cycles; "
        f"L1 accesses {mean.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.3g}",
        f"* vector ALU busy (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles)): {100 * 4.0 * mean.get('SQ_ACTIVE_INST_VALU', 0) / max(1.0, 256 * 4 * cyc):.0f} %",
        f"* waves {mean.get('SQ_WAVES', 0):.0f}; wave time waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES): "
        f"{100 * mean.get('SQ_WAIT_ANY', 0) / max(1.0, mean.get('SQ_WAVE_CYCLES', 0)):.0f} %",
        f"* VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): "
        f"{100 * mean.get('SQ_THREAD_CYCLES_VALU', 0) / max(1.0, 64 * mean.get('SQ_ACTIVE_INST_VALU', 0)):.0f} %; "
        f"VALU wave-instructions {mean.get('SQ_INSTS_VALU', 0):.3g}, SALU {mean.get('SQ_INSTS_SALU', 0):.3g}, "
        f"vector loads {mean.get('SQ_INSTS_VMEM_RD', 0):.3g}, LDS {mean.get('SQ_INSTS_LDS', 0):.3g}",
    ]
if bench:
    r = bench.get("roofline", {})
    ro, valu, hbm = r.get("ref_order", {}), r.get("valu", {}), r.get("hbm", {})
    lines += ["", f"bench.py line of the same build: {bench.get('value', 0):.1f} Mrays/s, {bench.get('ms_per_step', 0):.2f} ms/step, "
                  f"device ms per step {bench.get('config', {}).get('device_ms_per_step')}; "
                  f"{r.get('node_visits_per_ray', 0):.1f} node visits, {r.get('box_tests_per_ray', 0):.0f} box tests and {r.get('prim_tests_per_ray', 0):.2f} primitive tests per ray; "
                  f"bound: {r.get('bound')}; HBM roof on the bytes of the kernel's own visits {hbm.get('achieved', 0):.0f} GB/s = {hbm.get('frac', 0):.3f} of peak "
                  f"(on the bytes its formats request {hbm.get('frac_requested', 0):.3f}, on the measured fabric traffic {hbm.get('traffic_frac') or 0:.3f}); "
                  f"vector-ALU roof {valu.get('achieved', 0):.0f} G lane-instructions/s = {valu.get('frac', 0):.3f} of {valu.get('peak', 0):.0f} at {valu.get('clock_ghz', 0):.2f} GHz; "
                  f"the same formulas on the reference-order counts: HBM {ro.get('hbm_frac', 0):.2f}, vector ALU {ro.get('valu_frac', 0):.2f} of peak "
                  f"(visit reduction {ro.get('visit_reduction', 0):.1f}x); cpu_baseline "
                  f"{bench.get('cpu_baseline', {}).get('value', 0):.3f} Mrays/s on {bench.get('cpu_baseline', {}).get('cores', 0)} threads."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(l for l in lines if l is not None) + "\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
