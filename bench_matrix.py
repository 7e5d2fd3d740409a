#!/usr/bin/env python3
"""bench_matrix.py -- closest-hit throughput over the synthetic scenes and ray batches of the
measurement plan: random spheres (485 Book-1 generator, 1e4, 1e6), the 2^20-triangle soup, a
bunny-class closed mesh; batches (A) coherent primary rays, (B) incoherent rays, (C) secondary rays
(cosine-weighted bounces off the hits of A).  For every cell: kernel ms (HIP events), Mrays/s, hit
fraction, and parity of a 65,536-ray sample against the CPU oracle.  Not the driver's bench contract
(that is bench.py); a measurement tool whose output is kept under profiles/.

  python bench_matrix.py [--rays 16777216] [--out profiles/r1_matrix.md]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--sample", type=int, default=1 << 16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scenes", default="book1,spheres1e4,spheres1e6,soup,bunny,cornell,book2")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import numpy as np
    import torch

    import oracle_ffi as O
    import ray_tracer_amd as rt
    from ray_tracer_amd import _ffi as F
    from ray_tracer_amd import scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench_matrix.py needs a HIP device: the product has no CPU path")
    dev = torch.device("cuda", 0)
    n = args.rays
    side = int(math.isqrt(n))
    stream = torch.cuda.current_stream().cuda_stream

    def build(name):
        if name == "book1":
            world, _, _ = scenes.book1_final(seed=0)
            return world
        if name == "spheres1e4":
            return scenes.random_spheres(10_000, seed=1)
        if name == "spheres1e6":
            return scenes.random_spheres(1_000_000, seed=1)
        if name == "soup":
            return scenes.triangle_soup_scene(1 << 20, seed=2)
        if name == "cornell":  # general kernel: flat list, instance, glass sphere (main.rs:421-551)
            return scenes.cornell_box()[0]
        if name == "book2":    # general kernel: two BVHs, lists, instances, constant media (main.rs:651-788)
            return scenes.book2_final(seed=0)[0]
        if name == "bunny":
            return rt.Mesh(scenes.icosphere_mesh(6, seed=4), rt.Lambertian(rt.Color(0.5, 0.5, 0.5)))
        raise SystemExit(f"unknown scene {name}")

    def to_dev(o, d, tm):
        r = {}
        for k, nm in enumerate(("ox", "oy", "oz")):
            r[nm] = torch.from_numpy(np.ascontiguousarray(o[:, k])).to(dev)
        for k, nm in enumerate(("dx", "dy", "dz")):
            r[nm] = torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev)
        r["time"] = torch.from_numpy(np.ascontiguousarray(tm)).to(dev)
        return r

    rows = []
    for name in args.scenes.split(","):
        t0 = time.time()
        world = build(name)
        desc = rt.SceneDesc.from_world(world)
        gpu = rt.Scene(desc)
        cpu = O.OracleScene(desc)
        info = gpu.info()
        build_s = time.time() - t0
        b = [info.world_bounds[k] for k in range(6)]
        if name == "book1":  # look at the small spheres, not at the 1000-radius ground sphere's box
            b = [-11.0, 11.0, 0.0, 2.0, -11.0, 11.0]
        if name == "book2":  # the room, not the 5000-radius fog boundary
            b = [-1000.0, 1000.0, 0.0, 600.0, -1000.0, 1000.0]
        m = side * side
        hits = {"t": torch.empty(m, dtype=torch.float64, device=dev), "prim": torch.empty(m, dtype=torch.int32, device=dev)}
        for nm in ("u", "v", "nx", "ny", "nz", "px", "py", "pz"):
            hits[nm] = torch.empty(m, dtype=torch.float64, device=dev)
        hits["front_face"] = torch.empty(m, dtype=torch.uint8, device=dev)
        hits["mat"] = torch.empty(m, dtype=torch.int32, device=dev)
        batches = {}
        o, d, tm = scenes.primary_rays(side, b)
        if name == "cornell":  # look in through the open front (main.rs:536 puts the camera at z = -800), not at the back of the back wall
            o[:, 2] = (b[4] + b[5]) - o[:, 2]
            d[:, 2] = -d[:, 2]
        batches["A primary"] = to_dev(o, d, tm)
        o, d, tm = scenes.incoherent_rays(m, b, seed=3)
        batches["B incoherent"] = to_dev(o, d, tm)
        # (C): bounce the primary rays off what they hit, cosine-weighted about the shading normal
        gpu.closest_hit_device(batches["A primary"], hits, m, seed=0, flags=F.RT_HIT_ORDERED, stream=stream)
        torch.cuda.synchronize()
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        r1 = torch.rand(m, dtype=torch.float64, device=dev, generator=g)
        r2 = torch.rand(m, dtype=torch.float64, device=dev, generator=g)
        nrm = torch.stack([hits["nx"], hits["ny"], hits["nz"]], 1)
        hit = hits["prim"] != -1
        a = torch.where((nrm[:, 0].abs() > 0.9).unsqueeze(1), torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64, device=dev),
                        torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64, device=dev))
        v = torch.linalg.cross(nrm, a)
        v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-300)
        u = torch.linalg.cross(nrm, v)
        phi = 2.0 * math.pi * r1
        sd = u * (torch.cos(phi) * torch.sqrt(r2)).unsqueeze(1) + v * (torch.sin(phi) * torch.sqrt(r2)).unsqueeze(1) + nrm * torch.sqrt(1.0 - r2).unsqueeze(1)
        pa = batches["A primary"]
        sec = {"ox": torch.where(hit, hits["px"], pa["ox"]).contiguous(), "oy": torch.where(hit, hits["py"], pa["oy"]).contiguous(),
               "oz": torch.where(hit, hits["pz"], pa["oz"]).contiguous(), "dx": torch.where(hit, sd[:, 0], pa["dx"]).contiguous(),
               "dy": torch.where(hit, sd[:, 1], pa["dy"]).contiguous(), "dz": torch.where(hit, sd[:, 2], pa["dz"]).contiguous(),
               "time": pa["time"]}
        batches["C secondary"] = sec
        del nrm, v, u, sd, r1, r2, a

        for bname, rays in batches.items():
            gpu.closest_hit_device(rays, hits, m, seed=0, flags=F.RT_HIT_ORDERED, stream=stream)  # warm-up
            torch.cuda.synchronize()
            ms = []
            t1 = time.perf_counter()
            for _ in range(args.steps):
                gpu.closest_hit_device(rays, hits, m, seed=0, flags=F.RT_HIT_ORDERED, stream=stream)
                c_ = gpu.counters()
                ms.append(c_.ms_last_traversal)  # the traversal kernel alone; a reorder pass in front of it is reported beside it
                pre_ms = c_.ms_last_reorder
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t1) / args.steps
            hit_frac = float((hits["prim"] != -1).double().mean().item())
            # parity of a sample against the oracle (reference-order recursion): closest t must be identical
            k = min(args.sample, m)
            idx = torch.linspace(0, m - 1, k, device=dev).long()
            so = np.stack([rays[c][idx].cpu().numpy() for c in ("ox", "oy", "oz")], 1)
            sdv = np.stack([rays[c][idx].cpu().numpy() for c in ("dx", "dy", "dz")], 1)
            st = rays["time"][idx].cpu().numpy()
            ref = cpu.closest_hit(so, sdv, time=st, nthreads=min(16, os.cpu_count() or 1))
            if info.n_media:
                # a medium's scattering distance is drawn from the counter RNG keyed by the ray's index in its
                # batch: compare like with like by tracing the gathered sample as a batch of its own
                srays = {c: rays[c][idx].contiguous() for c in rays}
                shits = {c: torch.empty(k, dtype=hits[c].dtype, device=dev) for c in hits}
                gpu.closest_hit_device(srays, shits, k, seed=0, flags=F.RT_HIT_ORDERED, stream=stream)
                torch.cuda.synchronize()
                tg = shits["t"].cpu().numpy()
                pg = shits["prim"].cpu().numpy().astype(np.uint32)
            else:
                tg = hits["t"][idx].cpu().numpy()
                pg = hits["prim"][idx].cpu().numpy().astype(np.uint32)
            # a constant-medium hit goes through ln(U): device and glibc logarithms may differ in the last place
            t_same = int(np.sum((tg == ref["t"]) | (np.isinf(tg) & np.isinf(ref["t"])) |
                                (((ref["prim"] >> 28) == F.RT_REF_MEDIUM) & np.isclose(tg, ref["t"], rtol=1e-12, atol=0))))
            p_same = int(np.sum(pg == ref["prim"]))
            rows.append({"scene": name, "prims": int(len(desc.arrays["spheres"]) + len(desc.arrays["triangles"]) + len(desc.arrays["quads"])), "nodes": int(info.n_nodes),
                         "batch": bname, "rays": m, "kernel_ms": float(np.mean(ms)), "reorder_ms": float(pre_ms), "step_ms": wall * 1e3,
                         "Mrays_per_s": m / wall / 1e6, "Mrays_per_s_kernel": m / (float(np.mean(ms)) * 1e-3) / 1e6, "hit_fraction": hit_frac,
                         "sample": k, "t_identical": t_same, "prim_identical": p_same, "build_s": build_s})
            print(json.dumps(rows[-1]), flush=True)
        del gpu, cpu, batches, hits
        torch.cuda.empty_cache()

    if args.out:
        with open(args.out, "w") as fh:
            fh.write("# Closest-hit throughput matrix (one MI355X, ordered traversal, 2^24-class batches, f64 rays)\n\n")
            fh.write("| scene | primitives | BVH nodes | batch | traversal kernel ms | reorder pass ms | Mrays/s (kernel) | Mrays/s (step: reorder, traversal, record completion) | hit fraction | "
                     "sample vs oracle: t identical | prim identical |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
            for r in rows:
                fh.write(f"| {r['scene']} | {r['prims']} | {r['nodes']} | {r['batch']} | {r['kernel_ms']:.2f} | {r['reorder_ms']:.2f} | {r['Mrays_per_s_kernel']:.0f} | "
                         f"{r['Mrays_per_s']:.0f} | {r['hit_fraction']:.3f} | {r['t_identical']}/{r['sample']} | {r['prim_identical']}/{r['sample']} |\n")
            fh.write("\nprim differences at identical t are ties between primitives hit at exactly the same distance (ordered traversal may "
                     "report either; the reference order reports the later one in list order).\n")


if __name__ == "__main__":
    main()
