#!/bin/bash
# Parameter sweep of the primitive-BVH kernel on the bench workload (kernel ms per setting).
# Usage: profiles/sweep.sh "<VAR=val VAR=val>" ...   (one quoted group per run)
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 200 python bench.py --no-cpu --no-render --steps 3 2>/dev/null) || exit 1
  echo "$cfg  $(echo "$out" | grep -o '"kernel_ms": [0-9.]*')  $(echo "$out" | grep -o '"value": [0-9.]*')"
done
