#!/bin/bash
# Memory-path counter passes for bench.py (texture addresser / vector L1 / L2-to-fabric), one rocprofv3 run per pass.
# Usage: profiles/run_pmc_mem.sh <outdir>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-render --no-mesh > "$out/$name.log" 2>&1 || exit 1; }
mkdir -p "$out"
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum
