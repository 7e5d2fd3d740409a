#!/bin/bash
# Kernel-trace statistics and counter passes of one full render (bench_render.py), one rocprofv3 run per pass.
# Usage: profiles/r2_render_profile.sh <outdir> <scene> [bench_render args...]      (run on the GPU box)
out=$1; scene=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench_render.py --scene "$scene" --cpu-rows 0 "$@" > "$out/trace.log" 2>&1 || exit 1
run() { name=$1; shift; timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d "$out/$name" -- python3 bench_render.py --scene "$scene" --cpu-rows 0 "$@" > "$out/$name.log" 2>&1 || exit 1; }
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY" run sq1 "$@"
PMC="SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" run sq2 "$@"
PMC="FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE" run fetch "$@"
PMC="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" run write "$@"
PMC="TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" run ta "$@"
tail -1 "$out/trace.log" | cut -c1-600
