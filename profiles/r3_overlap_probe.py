#!/usr/bin/env python3
"""How much of the binning pass, the record completion and the end of a traversal launch can hide behind the traversal
of another part of the same batch?  The 2^24-ray batch of bench.py is cut into K pieces that go, alternating, to two
scenes (two copies of the same tree, so that the pieces share no scratch) on two streams; K = 1 is the product's
single call.  Prints the time per whole batch.

  python profiles/r3_overlap_probe.py [soup|mesh]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import _ffi as F  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "soup"
n = 1 << 24
dev = torch.device("cuda", 0)
if which == "soup":
    desc = rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 20, seed=2))
else:
    desc = rt.SceneDesc.from_world(rt.Mesh(scenes.icosphere_mesh(subdiv=8), rt.Lambertian(rt.Color(.5, .5, .5))))
pair = [rt.Scene(desc), rt.Scene(desc)]
b = pair[0].info().world_bounds
o, d, tm = scenes.incoherent_rays(n, [b[k] for k in range(6)], seed=3)
rays = {}
for k, name in enumerate(("ox", "oy", "oz")):
    rays[name] = torch.from_numpy(np.ascontiguousarray(o[:, k])).to(dev)
for k, name in enumerate(("dx", "dy", "dz")):
    rays[name] = torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev)
hits = {"t": torch.empty(n, dtype=torch.float64, device=dev), "prim": torch.empty(n, dtype=torch.int32, device=dev),
        "u": torch.empty(n, dtype=torch.float64, device=dev), "v": torch.empty(n, dtype=torch.float64, device=dev),
        "front_face": torch.empty(n, dtype=torch.uint8, device=dev), "mat": torch.empty(n, dtype=torch.int32, device=dev)}
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def run(pieces, reps=5):
    m = n // pieces
    views = []
    for j in range(pieces):
        views.append(({k: v[j * m:(j + 1) * m] for k, v in rays.items()}, {k: v[j * m:(j + 1) * m] for k, v in hits.items()}))
    best = 1e9
    for rep in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j, (r, h) in enumerate(views):
            pair[j % 2].closest_hit_device(r, h, m, seed=0, flags=F.RT_HIT_ORDERED, stream=streams[j % 2].cuda_stream)
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    return best * 1e3


ref_t = None
for pieces in (1, 2, 4, 8, 16):
    ms = run(pieces)
    chk = float(torch.nan_to_num(hits["t"], posinf=0.0).sum().item())
    if ref_t is None:
        ref_t = chk
    print(f"{which}: {pieces:2d} piece(s) on two scenes / two streams: {ms:.3f} ms per 2^24 rays = {n / ms / 1e3:.0f} Mrays/s; sum of t {'same' if chk == ref_t else 'DIFFERS'}", flush=True)
