#!/bin/bash
# A/B of two builds of the library on one box: traversal time of the two bench batches (soup, closed mesh), alternating, twice.
# Usage: profiles/r3_ab.sh <other library> [scenes...]
cd "$GRAFT_REPO_ROOT" || exit 1
other=$1; shift
scenes=${@:-soup mesh}
for rep in 1 2; do
  for scene in $scenes; do
    echo -n "base  $scene: "; PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-150 || exit 1
    echo -n "other $scene: "; MI355RT_LIB=$other PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-150 || exit 1
  done
done
