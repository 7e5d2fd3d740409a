"""RT_HIT_ORDERED | RT_HIT_EXACT_CULL against the oracle's BVH on rays aimed at the vertices and edge midpoints of a closed
mesh: how many closest t differ, by how much, with the f32 short cut on and off (MI355RT_EXACT_F32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import scenes, _ffi as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_ffi as O

m = rt.Lambertian(rt.Color(.5, .5, .5))
tris = scenes.icosphere_mesh(subdiv=4)
desc = rt.SceneDesc.from_world(rt.Mesh(tris, m))
bvh = O.OracleScene(desc)
g = np.random.default_rng(77)
verts = tris.reshape(-1, 3)
mids = 0.5 * (tris[:, 0] + tris[:, 1])
targets = np.concatenate([verts[g.integers(0, len(verts), 3000)], mids[g.integers(0, len(mids), 3000)]])
dirs = g.normal(size=targets.shape)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
for f32 in ("1", "0"):
    os.environ["MI355RT_EXACT_F32"] = f32
    gpu = rt.Scene(desc)
    for dist, unit in ((3.0, False), (1e3, True), (1e5, True), (1e5, False)):
        o = targets - dirs * dist
        d = dirs if unit else (targets - o)
        c = gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
        r = gpu.closest_hit(o, d, flags=F.RT_HIT_REFERENCE_ORDER)
        w = bvh.closest_hit(o, d, nthreads=8)
        bad = np.nonzero(c["t"].view(np.uint64) != w["t"].view(np.uint64))[0]
        badr = np.nonzero(r["t"].view(np.uint64) != w["t"].view(np.uint64))[0]
        rel = np.abs(c["t"][bad] - w["t"][bad]) / np.abs(w["t"][bad]) if len(bad) else np.zeros(1)
        fin = np.isfinite(rel)
        print(f"f32={f32} dist={dist} unit={unit}: exact-cull differs on {len(bad)} rays (reference-order mode: {len(badr)}); "
              f"closer {int((c['t'][bad] < w['t'][bad]).sum())}, farther {int((c['t'][bad] > w['t'][bad]).sum())}, "
              f"max finite relative difference {rel[fin].max() if fin.any() else 0:.3g}, hit/miss flips {int((~fin).sum())}", flush=True)
    gpu.close()

# the short cut must not change anything: same tree, same order, same culls, so every output is identical with it off
soup = rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 17, seed=2))
o, d, _ = scenes.incoherent_rays(1 << 21, [-1, 1, -1, 1, -1, 1], seed=5)
res = {}
for sure in ("1", "0"):
    os.environ["MI355RT_EXACT_F32"] = "1"
    os.environ["MI355RT_EXACT_SURE"] = sure
    for name, dsc, oo, dd in (("soup", soup, o, d), ("icosphere", desc, targets - dirs * 1e3, dirs)):
        gpu = rt.Scene(dsc)
        gpu.reset_counters()
        res[(name, sure)] = gpu.closest_hit(oo, dd, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
        gpu.close()
for name in ("soup", "icosphere"):
    a, b = res[(name, "1")], res[(name, "0")]
    print(name, "short cut on/off identical:", all(np.array_equal(a[f].view(np.uint8), b[f].view(np.uint8)) for f in a), flush=True)
