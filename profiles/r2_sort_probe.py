#!/usr/bin/env python3
"""What would ray reordering buy the primitive-BVH kernel?  The bench batch (2^24 incoherent rays against the
2^20-triangle soup) is permuted on the device before the launch: by the Morton code of the origin cell at
several grid resolutions, optionally with the direction octant as the major or minor key, and the kernel is
timed on each order (HIP events of the library).  Results are checked against the unsorted launch."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import _ffi as F  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402


def spread3(v):
    v = v.to(torch.int64)
    out = torch.zeros_like(v)
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def main():
    n = int(os.environ.get("PROBE_RAYS", 1 << 24))
    dev = torch.device("cuda", 0)
    if os.environ.get("PROBE_SCENE") == "mesh":  # the bench's second figure: closed surface of 1,310,720 triangles
        scene = rt.Scene(rt.SceneDesc.from_world(rt.Mesh(scenes.icosphere_mesh(subdiv=8), rt.Lambertian(rt.Color(.5, .5, .5)))))
        b = scene.info().world_bounds
        o, d, tm = scenes.incoherent_rays(n, [b[k] for k in range(6)], seed=5)
    else:
        scene = rt.Scene(rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 20, seed=2)))
        o, d, tm = scenes.incoherent_rays(n, [0, 1, 0, 1, 0, 1], seed=3)
    O = torch.from_numpy(o).to(dev)
    D = torch.from_numpy(d).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(perm, label):
        Os, Ds = (O, D) if perm is None else (O[perm], D[perm])
        rays = {nm: Os[:, k].contiguous() for k, nm in enumerate(("ox", "oy", "oz"))}
        rays.update({nm: Ds[:, k].contiguous() for k, nm in enumerate(("dx", "dy", "dz"))})
        hits = {"t": torch.empty(n, dtype=torch.float64, device=dev), "prim": torch.empty(n, dtype=torch.int32, device=dev)}
        ms = []
        for _ in range(4):
            scene.closest_hit_device(rays, hits, n, flags=F.RT_HIT_ORDERED | int(os.environ.get("PROBE_FLAGS", "0")), stream=stream)
            torch.cuda.synchronize()
            c = scene.counters()
            ms.append(c.ms_last_closest_hit)
            parts = (round(c.ms_last_reorder, 3), round(c.ms_last_traversal, 3))
            if os.environ.get("PROBE_FLAGS"):
                scene.reset_counters()
        t = hits["t"]
        check = (float(torch.where(torch.isfinite(t), t, torch.zeros_like(t)).sum().item()), int(hits["prim"].to(torch.int64).sum().item()))
        print(json.dumps({"order": label, "call_ms": round(min(ms[1:]), 3), "reorder_ms, traversal_ms (last)": parts, "Mrays/s": round(n / min(ms[1:]) / 1e3, 1),
                          "checksum (sum t, sum prim)": check}), flush=True)
        return hits

    only = os.environ.get("PROBE_ONLY")  # "none" = just the unsorted launch; "<bits>" = just that Morton order (for counter passes)
    if only == "none":
        run(None, "as generated")
        return
    if only:
        bits = int(only)
        cell = (O.clamp(0, 1 - 1e-9) * (1 << bits)).to(torch.int64)
        key = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
        run(torch.argsort(key, stable=True), f"morton {bits} bits/axis")
        return
    base = run(None, "as generated")
    octant = ((D[:, 0] < 0).to(torch.int64) | ((D[:, 1] < 0).to(torch.int64) << 1) | ((D[:, 2] < 0).to(torch.int64) << 2))
    for bits in (2, 3, 4, 5, 6, 7, 8):
        cell = (O.clamp(0, 1 - 1e-9) * (1 << bits)).to(torch.int64)
        key = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
        for label, k in ((f"morton {bits} bits/axis", key), (f"morton {bits} bits/axis, octant minor", key * 8 + octant),
                         (f"octant major, morton {bits} bits/axis", (octant << (3 * bits)) | key)):
            perm = torch.argsort(k, stable=True)
            h = run(perm, label)
            if bits == 4 and label.endswith("axis"):
                assert torch.equal(h["t"], base["t"][perm]), "sorted launch disagrees"
    scene.close()


if __name__ == "__main__":
    main()
