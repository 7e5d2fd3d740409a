#!/bin/bash
# RT_HIT_ORDERED | RT_HIT_EXACT_CULL on the soup: workgroups per CU, leaf quorum and refill threshold of the exact-cull variant
# (98 registers: four workgroups per CU can be resident).
cd "$GRAFT_REPO_ROOT" || exit 1
run() { local label=$1; shift; echo -n "$label: "; env "$@" PROBE_FLAGS=4 PROBE_SCENE=soup PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-140 || exit 1; }
run base X=1
for b in 3 4 5; do run blocks$b MI355RT_FAST_BLOCKS_PER_CU=$b; done
for v in 8 12 24; do run blocks4_vote$v MI355RT_FAST_BLOCKS_PER_CU=4 MI355RT_PREP_VOTE=$v; done
for r in 40 56 60; do run blocks4_refill$r MI355RT_FAST_BLOCKS_PER_CU=4 MI355RT_PREP_REFILL=$r; done
run blocks4_postpone1 MI355RT_FAST_BLOCKS_PER_CU=4 MI355RT_POSTPONE=1
run blocks4_postpone0 MI355RT_FAST_BLOCKS_PER_CU=4 MI355RT_POSTPONE=0
