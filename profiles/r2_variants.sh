#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "$@"; do
  echo -n "$lib reorder on : "; MI355RT_LIB=$GRAFT_REPO_ROOT/ray_tracer_amd/$lib.so PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
  echo -n "$lib reorder off: "; MI355RT_REORDER=0 MI355RT_LIB=$GRAFT_REPO_ROOT/ray_tracer_amd/$lib.so PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
done
