#!/bin/bash
# Scheduling values of the primitive-BVH kernel on the traversal-only tree: traversal time of the two bench batches
# (soup, closed mesh) per setting; profiles/r2_sort_probe.py drives one scene and one batch.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { # label, scene, env...
  local label=$1 scene=$2; shift 2
  echo -n "$label $scene: "
  env "$@" PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-150 || exit 1
}
for scene in soup mesh; do
  run base $scene X=1
  run refill48 $scene MI355RT_PREP_REFILL=48 MI355RT_REFILL=40
  run refill60 $scene MI355RT_PREP_REFILL=60 MI355RT_REFILL=56
  run vote12 $scene MI355RT_PREP_VOTE=12 MI355RT_VOTE=12
  run vote24 $scene MI355RT_PREP_VOTE=24 MI355RT_VOTE=24
  run postpone0 $scene MI355RT_POSTPONE=0
  run postpone1 $scene MI355RT_POSTPONE=1
  run blocks4 $scene MI355RT_FAST_BLOCKS_PER_CU=4
  run blocks5 $scene MI355RT_FAST_BLOCKS_PER_CU=5
  run reorder2 $scene MI355RT_REORDER=2
  run reorder0 $scene MI355RT_REORDER=0
done
