#!/bin/bash
# Two pending triangles per lane: traversal time against the leaf quorum (soup: prepared rays; closed mesh: as they come).
cd "$GRAFT_REPO_ROOT" || exit 1
run() { local label=$1 scene=$2; shift 2; echo -n "$label $scene: "; env "$@" PROBE_SCENE=$scene PROBE_ONLY=none timeout -k 5 150 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 | cut -c1-210 || exit 1; }
for v in 16 24 32 40 48; do run vote$v soup MI355RT_PREP_VOTE=$v; done
for v in 16 24 32 40; do run vote$v mesh MI355RT_VOTE=$v; done
run base soup X=1
run base mesh X=1
