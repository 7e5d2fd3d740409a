#!/bin/bash
# The five render figures of the design notes, GPU only (bench_render.py --cpu-rows 0), one JSON line each.
# Usage: profiles/render_set.sh [outfile]
out=${1:-/dev/stdout}
run() { timeout -k 10 200 python bench_render.py --cpu-rows 0 "$@" | tail -1 >> "$out" || exit 1; }
run --scene cornell --spp 100
run --scene book1 --width 1200 --spp 9
run --scene book2 --width 800 --spp 100
run --scene smoke --width 600 --spp 100
run --scene mesh_spheres --width 1024 --spp 64
