#!/bin/bash
# Refill threshold x leaf quorum x postponed tests on the closed 1.3 M-triangle mesh, batch taken as it comes.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$*: "; env "$@" PROBE_SCENE=mesh PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c66-100 || exit 1; }
for refill in 44 48 52 56; do for vote in 8 12 16 20; do for pp in 0 1; do run MI355RT_REFILL=$refill MI355RT_VOTE=$vote MI355RT_POSTPONE=$pp; done; done; done
