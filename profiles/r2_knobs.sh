#!/bin/bash
# Scheduling knobs of the primitive-BVH kernel on the Morton-sorted bench batch.
cd "$GRAFT_REPO_ROOT" || exit 1
for refill in 24 40 48 56 64; do for vote in 16 24 32; do
  echo -n "refill=$refill vote=$vote: "
  MI355RT_REFILL=$refill MI355RT_VOTE=$vote PROBE_ONLY=${ORDER:-5} timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
done; done
