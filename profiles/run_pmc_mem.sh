#!/bin/bash
# Memory-pipeline counter passes (texture addresser / L1) for bench.py.  Usage: profiles/run_pmc_mem.sh <outdir>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > "$out/$name.log" 2>&1 || exit 1; }
mkdir -p "$out"
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
