#!/bin/bash
# Traversal time against batch size (soup and closed mesh): what a launch carries that does not grow with the batch,
# and the launch timeline of the counting build (MI355RT_DEBUG with RT_HIT_COUNTERS: first wave out of work, last wave done).
cd "$GRAFT_REPO_ROOT" || exit 1
for scene in soup mesh; do
  for n in 4194304 8388608 16777216 33554432; do
    echo -n "$scene $n: "; PROBE_RAYS=$n PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 200 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-130 || exit 1
  done
  echo "$scene timeline:"; MI355RT_DEBUG=1 PROBE_FLAGS=2 PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 200 python3 profiles/r2_sort_probe.py 2>&1 | grep -i "timeline\|iterations\|lanes" | tail -4 | cut -c1-400
done
