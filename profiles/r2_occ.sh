#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 1 2 3 4 5; do
  echo -n "blocks/CU=$b reorder on : "; MI355RT_FAST_BLOCKS_PER_CU=$b PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
  echo -n "blocks/CU=$b reorder off: "; MI355RT_FAST_BLOCKS_PER_CU=$b MI355RT_REORDER=0 PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
done
