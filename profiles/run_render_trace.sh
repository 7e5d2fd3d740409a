#!/bin/bash
# Kernel-trace statistics of full renders (bench_render.py), one rocprofv3 run per scene.
# Usage: profiles/run_render_trace.sh <outdir> <scene> [bench_render args...]
out=$1; scene=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$scene" -- python3 bench_render.py --scene "$scene" --cpu-rows 0 "$@" > "$out/$scene.log" 2>&1 || exit 1
f=$(ls "$out/$scene"/*/*_kernel_stats.csv | head -1)
cut -d, -f1-4,8 "$f" | head -12
tail -1 "$out/$scene.log" | cut -c1-400
