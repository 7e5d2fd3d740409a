#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo -n "reorder on : "; PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
for refill in 40 48 56; do for vote in 12 16 24; do
  echo -n "reorder off refill=$refill vote=$vote: "
  MI355RT_REORDER=0 MI355RT_REFILL=$refill MI355RT_VOTE=$vote PROBE_ONLY=none timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
done; done
