#!/usr/bin/env python3
"""Tracked summary of one full-frame render profile (profiles/r2_render_profile.sh -> gpurun_out/<dir>): per kernel,
summed over the frame's launches, the kernel-trace time and the counters of the separate --pmc passes.
Usage: python profiles/r2_render_summary.py <gpurun_out subdir> <tag>      writes profiles/<tag>_{kernel_stats.csv,summary.md,counters.csv}"""
import csv
import glob
import json
import os
import shutil
import sys

sub, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", sub)
dst = os.path.join(root, "profiles")
KERNELS = ("raygen_kernel", "extend_kernel", "bvh4_closest_hit_kernel", "shade_kernel", "tail_kernel", "accumulate_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
time_ms, calls = {}, {}
for row in csv.DictReader(open(stats)):
    k = short(row["Name"])
    if k:
        time_ms[k] = time_ms.get(k, 0.0) + float(row["TotalDurationNs"]) / 1e6
        calls[k] = calls.get(k, 0) + int(row["Calls"])
tot = {}
passes = {}
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    passes.setdefault(os.path.dirname(f), []).append(f)
for f in [sorted(files, key=os.path.getmtime)[-1] for files in passes.values()]:  # only the last run of each pass
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            tot.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
with open(os.path.join(dst, f"{tag}_counters.csv"), "w") as fh:
    fh.write("kernel,counter,sum_over_frame\n")
    for k in tot:
        for c in sorted(tot[k]):
            fh.write(f"{k},{c},{tot[k][c]:.6g}\n")
run = json.loads([l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1])
rays = run["gpu_rays"]
lines = [f"# {tag}: one frame of `{run['scene']}` {run['width']}x{run['height']}, {run['strata_per_pixel']} samples/pixel, depth {run['max_depth']}",
         "",
         f"bench_render.py under rocprofv3 --kernel-trace: {run['paths']} camera paths, {rays} path-bounces (world.hit queries), device {run['gpu_device_ms']:.1f} ms, "
         f"{run['gpu_samples_per_s'] / 1e6:.0f} M samples/s wall.  Counters from separate --pmc passes of the same command (kernels run one at a time there).",
         "FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes); it counts Infinity-Cache hits, so it bounds HBM traffic from above.",
         "",
         "| kernel | launches | ms / frame | share | fabric GB / frame | B / path-bounce | TB/s while it runs | L2 hit | VALU busy | lane util. | wave wait |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
all_ms = sum(time_ms.values())
frame_bytes = 0.0
for k in KERNELS:
    if k not in time_ms:
        continue
    c = tot.get(k, {})
    fetch, write = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    frame_bytes += fetch + write
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the kernel's launches of the pass that carried it
    valu = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * cyc) if cyc else float("nan")
    lane = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64 * c.get("SQ_ACTIVE_INST_VALU", 0.0))
    wait = c.get("SQ_WAIT_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0))
    hit = c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_REQ_sum", 0.0)) if c.get("TCC_REQ_sum") else float("nan")
    lines.append(f"| {k} | {calls[k]} | {time_ms[k]:.2f} | {100 * time_ms[k] / all_ms:.0f} % | {(fetch + write) / 1e9:.2f} | {(fetch + write) / rays:.0f} | "
                 f"{(fetch + write) / (time_ms[k] * 1e-3) / 1e12:.2f} | {100 * hit:.0f} % | {100 * valu:.0f} % | {100 * lane:.0f} % | {100 * wait:.0f} % |")
lines += ["",
          f"Frame: {frame_bytes / 1e9:.1f} GB through the fabric = {frame_bytes / rays:.0f} bytes per path-bounce; at the 8 TB/s HBM peak that is "
          f"{frame_bytes / 8e12 * 1e3:.1f} ms of the {all_ms:.1f} ms the kernels take, i.e. a memory roofline fraction of {frame_bytes / 8e12 * 1e3 / all_ms:.2f}.",
          "VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 summed over the kernel's launches (SQ_ACTIVE_INST_VALU counts quad-cycles); lane utilisation = "
          "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wave wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
