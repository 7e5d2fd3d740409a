#!/bin/bash
# Reordered batches: scheduling knobs of the primitive-BVH kernel (bench batch as generated, binned by the library).
cd "$GRAFT_REPO_ROOT" || exit 1
echo -n "reorder off: "; MI355RT_REORDER=0 PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
for refill in 48 56 60 62; do for vote in 8 12 16 24; do
  echo -n "prep_refill=$refill prep_vote=$vote: "
  MI355RT_PREP_REFILL=$refill MI355RT_PREP_VOTE=$vote PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
done; done
