#!/bin/bash
# Memory-side counter passes for bench.py under the current environment (one rocprofv3 run per pass).
# Usage: profiles/run_pmc_mem.sh <outdir>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu > "$out/$name.log" 2>&1 || exit 1; }
mkdir -p "$out"
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
