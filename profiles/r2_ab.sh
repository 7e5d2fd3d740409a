#!/bin/bash
# A/B of library builds in one call on one box: traversal time of the two bench batches (soup, closed mesh) for every
# library under ray_tracer_amd/variants/, twice over, alternating.  Extra environment is passed through; AB_ENV holds
# settings that are applied to the libraries whose name contains AB_MATCH only.
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do for lib in ray_tracer_amd/variants/*.so; do for scene in soup mesh; do
  echo -n "$(basename $lib) $scene: "
  MI355RT_LIB=$PWD/$lib PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-110 || exit 1
done; done; done
