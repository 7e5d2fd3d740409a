#!/bin/bash
# Ablation builds of the library (extra vector-ALU or vector-memory instructions per node step) timed on the bench batch,
# as generated and Morton-sorted.  Usage: profiles/r2_ablate.sh  (the variant libraries are built beforehand, in-tree)
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in libmi355rt libmi355rt_valu24 libmi355rt_valu48 libmi355rt_vmem1 libmi355rt_vmem2; do
  for order in none 5; do
    echo -n "$lib order=$order: "
    MI355RT_LIB=$GRAFT_REPO_ROOT/ray_tracer_amd/$lib.so PROBE_ONLY=$order timeout -k 5 120 python3 profiles/r2_sort_probe.py 2>&1 | tail -1 || exit 1
  done
done
