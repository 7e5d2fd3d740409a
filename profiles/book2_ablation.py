#!/usr/bin/env python3
"""Where the time of the Book-2 final scene goes: renders it with single ingredients left out
(device ms per frame, 800x800, 16 spp, depth 40).  A measurement tool, not a test."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt
from ray_tracer_amd import scenes
from ray_tracer_amd.hittable import HittableList, ConstantMedium, Translate, BvhNode


def variant(drop):
    world, lights, cam = scenes.book2_final(image_width=800, samples_per_pixel=16, max_depth=40)
    out = HittableList()
    for k, o in enumerate(world.objects):
        is_medium = isinstance(o, ConstantMedium)
        if drop == "fog" and is_medium and o.density < 0.001: continue
        if drop == "media" and is_medium: continue
        if drop == "sphere_cloud" and isinstance(o, Translate): continue
        if drop == "boxes" and isinstance(o, BvhNode): continue
        out.add(o)
    return out, lights, cam


for drop in ("nothing", "fog", "media", "sphere_cloud", "boxes"):
    world, lights, cam = variant(drop)
    gpu = rt.Scene(rt.SceneDesc.from_world(world, lights))
    c = cam.to_struct()
    gpu.render(c, seed=1)
    gpu.reset_counters()
    gpu.render(c, seed=1)
    ctr = gpu.counters()
    print(f"without {drop:13s} device_ms {ctr.ms_last_render:7.1f}  rays {ctr.render_rays/1e6:7.1f} M  Mrays/s {ctr.render_rays/ctr.ms_last_render/1e3:7.0f}", flush=True)
