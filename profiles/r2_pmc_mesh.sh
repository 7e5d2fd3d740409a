#!/bin/bash
# Counter passes of the primitive-BVH kernel on the closed-mesh batch (bench.py's second figure), one rocprofv3 run per pass.
# Usage: profiles/r2_pmc_mesh.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export PROBE_SCENE=mesh PROBE_ONLY=none
run() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 profiles/r2_sort_probe.py > "$out/$name.log" 2>&1 || exit 1; }
mkdir -p "$out"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo collected
