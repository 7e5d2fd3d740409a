#!/usr/bin/env python3
"""Scene creation time of the benchmark mesh (2^20 triangles) and of a 2^22-triangle soup: BVH built by the host
recursion (MI355RT_DEVICE_BVH_MIN=0) and on the device (default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt
from ray_tracer_amd import scenes
for n in (1 << 20, 1 << 22):
    desc = rt.SceneDesc.from_world(scenes.triangle_soup_scene(n, seed=2))
    for label, env in (("host recursion", "0"), ("device build", None), ("device build", None)):
        if env is None:
            os.environ.pop("MI355RT_DEVICE_BVH_MIN", None)
        else:
            os.environ["MI355RT_DEVICE_BVH_MIN"] = env
        t0 = time.perf_counter()
        s = rt.Scene(desc)
        dt = time.perf_counter() - t0
        s.close()
        print(f"{n} triangles, {label}: rt_scene_create {dt:.3f} s", flush=True)
