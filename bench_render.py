#!/usr/bin/env python3
"""bench_render.py -- samples/s of full renders (the wavefront path tracer behind rt_render) for the
reference's scenes, next to the CPU oracle's recursive renderer on a bounded work unit.
Not the driver's bench contract (that is bench.py); a measurement tool whose output is kept
under profiles/.

  python bench_render.py [--scene cornell|book1|book2|smoke] [--spp N] [--width W]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--cpu-rows", type=int, default=8, help="image rows the oracle renders for the CPU figure")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import numpy as np

    import oracle_ffi as O
    import ray_tracer_amd as rt
    from ray_tracer_amd import _ffi as F
    from ray_tracer_amd import scenes

    build = {"cornell": scenes.cornell_box, "book1": scenes.book1_final, "book2": scenes.book2_final, "mesh_spheres": scenes.mesh_and_spheres,
             "smoke": scenes.cornell_smoke}[args.scene]
    world, lights, cam = build()
    if args.spp:
        cam.samples_per_pixel = args.spp
    if args.width:
        cam.image_width = args.width
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    c = cam.to_struct()
    d = cam.derived()
    n_strata = d.sqrt_spp * d.sqrt_spp
    paths = cam.image_width * d.image_height * n_strata
    gpu.render(c, seed=1, work=F.rt_work(0, 0, cam.image_width, min(8, d.image_height), 0, 1))  # warm-up: workspace + code
    gpu.reset_counters()
    t0 = time.perf_counter()
    accum = gpu.render(c, seed=1)
    wall = time.perf_counter() - t0
    ctr = gpu.counters()
    out = {"scene": args.scene, "width": cam.image_width, "height": d.image_height, "strata_per_pixel": n_strata,
           "max_depth": cam.max_depth, "paths": paths, "gpu_wall_s": wall, "gpu_device_ms": ctr.ms_last_render,
           "gpu_samples_per_s": paths / wall, "gpu_rays": int(ctr.render_rays), "gpu_Mrays_per_s": ctr.render_rays / wall / 1e6}
    if args.cpu_rows <= 0:  # GPU figure only (profiling runs)
        print(json.dumps(out))
        return
    # CPU oracle on a few rows of the same frame, all strata
    cpu = O.OracleScene(desc)
    rows = min(args.cpu_rows, d.image_height)
    y0 = d.image_height // 2 - rows // 2
    threads = min(16, os.cpu_count() or 1)
    stats = {}
    acc_cpu = np.zeros_like(accum)
    t0 = time.perf_counter()
    cpu.render(c, seed=1, work=F.rt_work(0, y0, cam.image_width, y0 + rows, 0, n_strata), accum=acc_cpu, nthreads=threads, stats=stats)
    cpu_s = time.perf_counter() - t0
    cpu_paths = cam.image_width * rows * n_strata
    sa = np.nan_to_num(accum[y0:y0 + rows] * d.pixel_samples_scale, posinf=1e6)
    sb = np.nan_to_num(acc_cpu[y0:y0 + rows] * d.pixel_samples_scale, posinf=1e6)
    out.update({"cpu_threads": threads, "cpu_rows": rows, "cpu_s": cpu_s, "cpu_samples_per_s": cpu_paths / cpu_s,
                "cpu_Mrays_per_s": stats["rays"] / cpu_s / 1e6, "rmse_vs_oracle_rows": float(np.sqrt(np.mean((sa - sb) ** 2))),
                "speedup_samples": (paths / wall) / (cpu_paths / cpu_s)})
    print(json.dumps(out))
    if args.out:
        rt.Image(cam, accum).write_ppm(args.out)


if __name__ == "__main__":
    main()
