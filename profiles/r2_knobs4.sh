#!/bin/bash
# Knobs of the primitive-BVH kernel once more, with 64-byte prepared records: refill threshold, leaf quorum, workgroups per CU.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$*: "; env "$@" PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-100 || exit 1; }
for scene in soup mesh; do
  run PROBE_SCENE=$scene
  for refill in 48 60 62; do run PROBE_SCENE=$scene MI355RT_PREP_REFILL=$refill; done
  for vote in 12 20; do run PROBE_SCENE=$scene MI355RT_PREP_VOTE=$vote; done
  for per_cu in 4 5 6; do run PROBE_SCENE=$scene MI355RT_FAST_BLOCKS_PER_CU=$per_cu; done
  for pp in 0 1; do run PROBE_SCENE=$scene MI355RT_POSTPONE=$pp; done
done
