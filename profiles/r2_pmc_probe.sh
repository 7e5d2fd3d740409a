#!/bin/bash
# Counter passes over profiles/r2_sort_probe.py (one rocprofv3 run per pass).  Usage: profiles/r2_pmc_probe.sh <outdir> <PROBE_ONLY value>
out=$1; export PROBE_ONLY=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 profiles/r2_sort_probe.py > "$out/$name.log" 2>&1 || exit 1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr
run fetch FETCH_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
