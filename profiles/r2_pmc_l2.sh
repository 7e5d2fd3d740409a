#!/bin/bash
# L2 hit rate of the primitive-BVH kernel for a library build: profiles/r2_pmc_l2.sh <outdir> <library> [soup|mesh]
out=$1; lib=$2; scene=${3:-soup}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI355RT_LIB=$PWD/$lib PROBE_SCENE=$scene PROBE_ONLY=none
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$out/tcc1" -- python3 profiles/r2_sort_probe.py > "$out/tcc1.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys
v = {}
for f in glob.glob(sys.argv[1] + "/tcc1/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "bvh4_closest_hit_kernel" in row["Kernel_Name"]:
            v.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in v.items()}
print(sys.argv[1], {k: f"{x:.4g}" for k, x in m.items()}, "L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
PY
