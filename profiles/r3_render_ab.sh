#!/bin/bash
# Camera samples/s of the three frame configurations with the library in the tree (bench_render.py, a few oracle rows each for the RMSE).
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "cornell 1000" "cornell 100" "book2 100" "book1 10" "smoke 100" "mesh_spheres 64"; do
  set -- $spec
  extra=""; [ "$1" = "mesh_spheres" ] && extra="--width 1024"
  timeout -k 5 200 python3 bench_render.py --scene $1 --spp $2 --cpu-rows 4 $extra 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-13s spp %5d: %7.1f M samples/s, device %.1f ms, rmse %.2e' % (j['scene'], j['strata_per_pixel'], j['gpu_samples_per_s']/1e6, j['gpu_device_ms'], j['rmse_vs_oracle_rows']))" || exit 1
done
