#!/bin/bash
# Refill threshold x leaf quorum of the prepared-ray variant at six waves per SIMD, 2^20-triangle soup.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$*: "; env "$@" PROBE_SCENE=soup PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c66-100 || exit 1; }
for refill in 52 56 60; do for vote in 12 16 20 24; do run MI355RT_PREP_REFILL=$refill MI355RT_PREP_VOTE=$vote; done; done
