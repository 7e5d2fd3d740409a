#!/bin/bash
# Samples/s and device ms of three frames for several values of MI355RT_TAIL_PATHS (live paths at which one launch
# finishes a batch).  Measured with the hoisted BVHs: 8192 / 32768 / 65536 / 262144 -> Book-1 690 / 748 / 766 / 705,
# mesh + spheres 967 / 1005 / 1026 / 946, Book-2 472 / 472 / 472 / 473 M samples/s.
for t in 8192 32768 65536 262144; do
  for sc in "book1 --width 1200 --spp 9" "mesh_spheres --width 1024 --spp 64" "book2 --width 800 --spp 100"; do
    echo -n "tail $t $sc: "
    MI355RT_TAIL_PATHS=$t timeout -k 10 200 python bench_render.py --cpu-rows 0 --scene $sc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['gpu_samples_per_s']/1e6,1), round(d['gpu_device_ms'],2))" || exit 1
  done
done
