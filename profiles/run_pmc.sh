#!/bin/bash
# Counter passes for bench.py (one rocprofv3 run per pass, counters only).  Usage: profiles/run_pmc.sh <outdir> [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-render --no-mesh $BENCH_ARGS > "$out/$name.log" 2>&1 || exit 1; }
mkdir -p "$out"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
