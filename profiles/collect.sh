#!/bin/bash
# Everything the tracked summaries of one round are made from, in one GPU call:
#   bench.py line, rocprofv3 kernel trace of the same command, counter passes (separate runs, counters only).
# Usage (on the GPU box, from the repository root): profiles/collect.sh <tag>
# Afterwards, here: python profiles/summarize.py <tag> bvh4_closest_hit_kernel
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > "gpurun_out/${tag}_bench.json" 2> "gpurun_out/${tag}_bench.err" || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/${tag}_trace" -- python3 bench.py --no-cpu --no-render --no-mesh > "gpurun_out/${tag}_trace.log" 2>&1 || exit 1
profiles/run_pmc.sh "gpurun_out/${tag}_pmc" || exit 1
profiles/run_pmc_mem.sh "gpurun_out/${tag}_pmc_mem" || exit 1
echo collected
