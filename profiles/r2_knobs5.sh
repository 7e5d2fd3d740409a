#!/bin/bash
# Knobs of the primitive-BVH kernel on batches it takes as they come (no reorder pass): the closed 1.3 M-triangle mesh.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$*: "; env "$@" PROBE_SCENE=mesh PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-100 || exit 1; }
run MI355RT_REORDER=1
for refill in 32 48 56; do run MI355RT_REFILL=$refill; done
for vote in 12 16 32; do run MI355RT_VOTE=$vote; done
for pp in 0 1; do run MI355RT_POSTPONE=$pp; done
for per_cu in 3 5; do run MI355RT_FAST_BLOCKS_PER_CU=$per_cu; done
for chunk in 32 64; do run MI355RT_CHUNK=$chunk; done
run MI355RT_REFILL=48 MI355RT_VOTE=16
