#!/usr/bin/env python3
"""bench.py -- closest-hit throughput of the HIP path on the 1M-triangle workload.

One step = one rt_closest_hit_device launch over one batch of 2^24 incoherent f64 rays
(origins uniform in the scene box, directions uniform on the sphere) against the
2^20-triangle soup held as a Mesh (BVH in the reference's split order).  Rays and hit
records are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank traces its own batch (weak scaling, no data-path collective); the
time is the max over ranks between two barriers.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
VALU_PEAK_TINSTR = 256 * 64 * 2.4e9 / 1e12  # lane-instructions per second, in units of 1e12
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 64, 72, 72, 48


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--triangles", type=int, default=1 << 20)
    ap.add_argument("--mode", choices=["reference", "ordered"], default=os.environ.get("MI355RT_BENCH_MODE", "ordered"))
    ap.add_argument("--cpu-sample", type=int, default=1 << 23, help="rays of the batch timed on the host cores")
    ap.add_argument("--count-sample", type=int, default=1 << 16, help="rays used for the reference-order visit counts")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-render", action="store_true", help="skip the camera-samples/s figure (second half of the metric)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import ray_tracer_amd as rt
    from ray_tracer_amd import _ffi as F
    from ray_tracer_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    # ------------------------------------------------------------------ scene + batch
    t0 = time.time()
    mesh = scenes.triangle_soup_scene(args.triangles, seed=2)
    desc = rt.SceneDesc.from_world(mesh)
    scene = rt.Scene(desc)
    info = scene.info()
    build_s = time.time() - t0
    bounds = [info.world_bounds[k] for k in range(6)]
    o, d, tm = scenes.incoherent_rays(args.rays, bounds, seed=3 + rank)
    rays = {}
    for k, name in enumerate(("ox", "oy", "oz")):
        rays[name] = torch.from_numpy(np.ascontiguousarray(o[:, k])).to(dev)
    for k, name in enumerate(("dx", "dy", "dz")):
        rays[name] = torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev)
    rays["time"] = torch.from_numpy(tm).to(dev)
    n = args.rays
    hits = {"t": torch.empty(n, dtype=torch.float64, device=dev), "prim": torch.empty(n, dtype=torch.int32, device=dev),
            "u": torch.empty(n, dtype=torch.float64, device=dev), "v": torch.empty(n, dtype=torch.float64, device=dev),
            "front_face": torch.empty(n, dtype=torch.uint8, device=dev), "mat": torch.empty(n, dtype=torch.int32, device=dev)}
    flags = F.RT_HIT_ORDERED if args.mode == "ordered" else F.RT_HIT_REFERENCE_ORDER
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        scene.closest_hit_device(rays, hits, n, seed=0, flags=flags, stream=stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        # device time of this launch from HIP events recorded on the launch stream
        kernel_ms.append(scene.counters().ms_last_closest_hit)
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # one extra launch outside the timed region with the visit counters on: what this traversal actually did
    scene.reset_counters()
    scene.closest_hit_device(rays, hits, n, seed=0, flags=flags | F.RT_HIT_COUNTERS, stream=stream)
    torch.cuda.synchronize()
    cnt = scene.counters()
    actual_nodes, actual_prims = cnt.node_visits / n, cnt.prim_tests / n
    total_rays = float(n) * args.steps * world_size
    mrays = total_rays / elapsed / 1e6
    hit_frac = float((hits["prim"] != -1).double().mean().item())

    out = {
        "metric": "Mrays/s closest-hit", "value": mrays, "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"closest-hit, {args.triangles}-triangle soup Mesh (BVH, reference split order), "
                               f"{n} incoherent f64 rays per GPU per step", "traversal": args.mode,
                   "bvh_nodes": int(info.n_nodes), "scene_bytes": int(info.device_bytes), "hit_fraction": hit_frac,
                   "scene_build_s": build_s},
    }

    if rank == 0:
        import oracle_ffi as O  # the CPU oracle: checker and reported baseline only
        cpu = O.OracleScene(desc)
        # reference-order visit counts per ray (define the algorithmic bytes of a launch)
        m = min(args.count_sample, n)
        ctr = {}
        cpu.closest_hit(o[:m], d[:m], time=tm[:m], nthreads=min(16, os.cpu_count() or 1), counters=ctr)
        nodes_per_ray = ctr["node_visits"] / m
        prims_per_ray = ctr["prim_tests"] / m
        bytes_per_ray = NODE_BYTES * nodes_per_ray + TRI_BYTES * prims_per_ray + RAY_BYTES + HIT_BYTES
        avg_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_ray * n / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": _traffic_from_env(),
                           "kernel": "bvh4_closest_hit_kernel" if args.mode == "ordered" else "closest_hit_kernel", "kernel_ms": avg_ms,
                           "ref_order_nodes_per_ray": nodes_per_ray, "ref_order_prims_per_ray": prims_per_ray,
                           "alg_bytes_per_ray": bytes_per_ray, "count_sample_rays": m,
                           "actual_box_tests_per_ray": actual_nodes, "actual_prim_tests_per_ray": actual_prims,
                           # ordered: 64-byte 4-wide nodes (16 B per child box tested), one 128-byte line per triangle
                           "actual_bytes_per_ray": (16 * actual_nodes + 128 * actual_prims if args.mode == "ordered"
                                                    else 64 * actual_nodes + 128 * actual_prims) + RAY_BYTES + HIT_BYTES}
        out["roofline"]["achieved_actual"] = out["roofline"]["actual_bytes_per_ray"] * n / (avg_ms * 1e-3) / 1e9
        # the second candidate roof: vector ALU lane-instructions (256 CUs x 64 lanes x 2.4 GHz), with the same
        # reference-order visit counts: 45 per node visit (f64 slab test), 70 per triangle test, 40 per ray
        instr_per_ray = 45.0 * nodes_per_ray + 70.0 * prims_per_ray + 40.0
        valu_achieved = instr_per_ray * n / (avg_ms * 1e-3) / 1e12
        out["roofline"]["valu"] = {"alg_lane_instr_per_ray": instr_per_ray, "achieved": valu_achieved, "peak": VALU_PEAK_TINSTR,
                                   "unit": "T lane-instr/s", "frac": valu_achieved / VALU_PEAK_TINSTR}
        if out["roofline"]["traffic"]:
            # measured L2 <-> fabric bytes per launch (profiles/) over this run's kernel time: the binding roof
            out["roofline"]["traffic_gbs"] = out["roofline"]["traffic"] / (avg_ms * 1e-3) / 1e9
            out["roofline"]["traffic_frac"] = out["roofline"]["traffic_gbs"] / HBM_PEAK_GBS
        if not args.no_cpu and world_size == 1:
            threads = min(16, os.cpu_count() or 1)
            mc = min(args.cpu_sample, n)
            t1 = time.perf_counter()
            ref = cpu.closest_hit(o[:mc], d[:mc], time=tm[:mc], nthreads=threads)
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": mc / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": f"first {mc} rays of the same batch, C++ f64 restatement of the reference "
                                             f"(reference-order traversal), {threads} threads, {cpu_s:.1f} s"}
            # the same restatement on one core, on a smaller slice of the batch
            m1 = min(mc, max(1, mc // 16))
            t1 = time.perf_counter()
            cpu.closest_hit(o[:m1], d[:m1], time=tm[:m1], nthreads=1)
            one_s = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": m1 / one_s / 1e6, "unit": "Mrays/s", "sample": f"first {m1} rays, {one_s:.1f} s"}
            got = hits["prim"][:mc].cpu().numpy().astype(np.uint32)
            same = got == ref["prim"]
            t_gpu = hits["t"][:mc].cpu().numpy()
            tie = np.isclose(t_gpu, ref["t"], rtol=1e-6, atol=0) | (np.isinf(t_gpu) & np.isinf(ref["t"]))
            out["config"]["parity_vs_oracle"] = {"rays": int(mc), "prim_mismatch_not_tie": int((~same & ~tie).sum()),
                                                 "prim_mismatch_tie": int((~same & tie).sum())}
    if rank == 0 and world_size == 1 and not args.no_render:
        out["render"] = {name: _render_figure(rt, F, scenes, np, name, with_cpu=not args.no_cpu) for name in ("book2_final", "cornell_box")}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _render_figure(rt, F, scenes, np, name, with_cpu):
    """Second half of the metric: camera samples/s of a full frame next to the oracle on some rows of it.
    book2_final: final_scene (main.rs:651-788), 800x800, depth 40 (BVHs, instances, media, textures);
    cornell_box: the scene live at the reference's HEAD (main.rs:421-551), 600x600, depth 50, mixture PDF.
    100 samples per pixel each (the reference's 10000 / 1000 only lengthen the run)."""
    if name == "book2_final":
        world, lights, cam = scenes.book2_final(image_width=800, samples_per_pixel=100, max_depth=40)
    else:
        world, lights, cam = scenes.cornell_box(image_width=600, samples_per_pixel=100, max_depth=50)
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    c, d = cam.to_struct(), cam.derived()
    n_strata = d.sqrt_spp * d.sqrt_spp
    paths = cam.image_width * d.image_height * n_strata
    gpu.render(c, seed=1, work=F.rt_work(0, 0, cam.image_width, 8, 0, 1))  # workspace + code
    t0 = time.perf_counter()
    accum = gpu.render(c, seed=1)
    wall = time.perf_counter() - t0
    fig = {"metric": "camera samples/s", "scene": f"{name} {cam.image_width}x{d.image_height}, {n_strata} samples/pixel, depth {cam.max_depth}",
           "value": paths / wall, "unit": "samples/s", "ms_frame": 1e3 * wall, "device_ms": gpu.counters().ms_last_render}
    if with_cpu:
        import oracle_ffi as O
        rows, threads = (200 if name == "book2_final" else 600), min(16, os.cpu_count() or 1)
        y0 = max(0, (d.image_height - rows) // 2)
        acc = np.zeros_like(accum)
        t0 = time.perf_counter()
        O.OracleScene(desc).render(c, seed=1, work=F.rt_work(0, y0, cam.image_width, y0 + rows, 0, n_strata), accum=acc, nthreads=threads)
        cpu_s = time.perf_counter() - t0
        sa = np.nan_to_num(accum[y0:y0 + rows] * d.pixel_samples_scale, posinf=1e6)
        sb = np.nan_to_num(acc[y0:y0 + rows] * d.pixel_samples_scale, posinf=1e6)
        fig["cpu_baseline"] = {"value": cam.image_width * rows * n_strata / cpu_s, "unit": "samples/s", "cores": threads, "kind": "port",
                               "sample": f"{rows} image rows of the same frame, all strata, {cpu_s:.1f} s"}
        fig["rmse_vs_oracle_rows"] = float(np.sqrt(np.mean((sa - sb) ** 2)))
    return fig


def _traffic_from_env():
    """HBM bytes per launch from a separate rocprofv3 --pmc pass, when one was recorded (profiles/)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    main()
