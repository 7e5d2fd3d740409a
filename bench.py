#!/usr/bin/env python3
"""bench.py -- closest-hit throughput of the HIP path on the 1M-triangle workload.

One step = one rt_closest_hit_device call over one batch of 2^24 incoherent f64 rays
(origins uniform in the scene box, directions uniform on the sphere) against the
2^20-triangle soup held as a Mesh (BVH in the reference's split order): the library's
binning pass, the traversal kernel and the pass that completes the hit records.  Rays and
hit records are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank traces its own batch (weak scaling, no data-path collective); the
time is the max over ranks between two barriers.  After that leg an N-rank job renders a
bounded sample of the 4096 x 2304 mesh + spheres frame: every rank its band of the strata
on its own GPU, then ONE sum-reduction of the f64 accumulator to rank 0 over RCCL
(ray_tracer_amd.parallel.render_frame_job); reported under "multi_gpu_render".
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 64, 72, 72, 48  # record sizes of the algorithmic-bytes formula


def source_hash() -> str:
    """Hash of the kernel sources: a profile under profiles/ only describes the build it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ray_tracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


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
    ap.add_argument("--no-render", action="store_true", help="skip the camera-samples/s figures (second half of the metric)")
    ap.add_argument("--no-mesh", action="store_true", help="skip the closed-mesh closest-hit figure")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) and run the frame job even with one rank")
    ap.add_argument("--frame-strata", type=int, default=64, help="strata per rank of the multi-GPU frame sample (of 4096)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import ray_tracer_amd as rt
    from ray_tracer_amd import _ffi as F
    from ray_tracer_amd import parallel, scenes

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local_rank)  # before any scene is created: a scene lives on the device current at rt_scene_create
    dev = torch.device("cuda", local_rank)
    dist = None
    if world_size > 1 or args.force_dist:  # --force-dist: rehearse the RCCL path with a single rank under torch.distributed.run
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    flags = F.RT_HIT_ORDERED if args.mode == "ordered" else F.RT_HIT_REFERENCE_ORDER
    n = args.rays

    def device_rays(o, d, tm):
        rays = {}
        for k, name in enumerate(("ox", "oy", "oz")):
            rays[name] = torch.from_numpy(np.ascontiguousarray(o[:, k])).to(dev)
        for k, name in enumerate(("dx", "dy", "dz")):
            rays[name] = torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev)
        rays["time"] = torch.from_numpy(tm).to(dev)
        return rays

    hits = {"t": torch.empty(n, dtype=torch.float64, device=dev), "prim": torch.empty(n, dtype=torch.int32, device=dev),
            "u": torch.empty(n, dtype=torch.float64, device=dev), "v": torch.empty(n, dtype=torch.float64, device=dev),
            "front_face": torch.empty(n, dtype=torch.uint8, device=dev), "mat": torch.empty(n, dtype=torch.int32, device=dev)}

    def timed_steps(scene, rays, steps, warmup):
        """`steps` calls between two barriers; device times from the library's HIP events on the launch stream."""
        for _ in range(warmup):
            scene.closest_hit_device(rays, hits, n, seed=0, flags=flags, stream=stream)
        barrier()
        scene.reset_counters()
        t_start = time.perf_counter()
        for _ in range(steps):
            scene.closest_hit_device(rays, hits, n, seed=0, flags=flags, stream=stream)
        barrier()
        elapsed = time.perf_counter() - t_start
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        c = scene.counters()  # harvests the events of the calls above (at most 32 are kept)
        k = max(1, c.timed_calls)
        return elapsed, {"traversal_ms": c.ms_sum_traversal / k, "reorder_ms": c.ms_sum_reorder / k, "call_ms": c.ms_sum_closest_hit / k,
                         "timed_calls": int(c.timed_calls)}

    def actual_counts(scene, rays):
        """One launch outside the timed region with the visit counters on: what this traversal actually did."""
        scene.reset_counters()
        scene.closest_hit_device(rays, hits, n, seed=0, flags=flags | F.RT_HIT_COUNTERS, stream=stream)
        torch.cuda.synchronize()
        c = scene.counters()
        return c.node_visits / n, c.prim_tests / n, int(c.shader_khz)

    # ------------------------------------------------------------------ headline: the 2^20-triangle soup
    # scene_build_s is rt_scene_create alone (validate, build the trees, upload).  The process's one-time costs (HIP
    # context, loading the library's code object, the first launches) are paid by a scene of a few spheres first, and the
    # numpy generation of the triangles is not part of it either; both are reported beside it.
    t0 = time.time()
    rt.Scene(rt.SceneDesc.from_world(scenes.triangle_soup_scene(2048, seed=1))).close()
    init_s = time.time() - t0
    t0 = time.time()
    desc = rt.SceneDesc.from_world(scenes.triangle_soup_scene(args.triangles, seed=2))
    describe_s = time.time() - t0
    t0 = time.time()
    scene = rt.Scene(desc)
    build_s = time.time() - t0
    info = scene.info()
    bounds = [info.world_bounds[k] for k in range(6)]
    o, d, tm = scenes.incoherent_rays(n, bounds, seed=3 + rank)
    rays = device_rays(o, d, tm)
    elapsed, dev_ms = timed_steps(scene, rays, args.steps, args.warmup)
    box_tests, prim_tests, shader_khz = actual_counts(scene, rays)
    mrays = float(n) * args.steps * world_size / elapsed / 1e6
    hit_frac = float((hits["prim"] != -1).double().mean().item())

    out = {
        "metric": "Mrays/s closest-hit", "value": mrays, "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"closest-hit, {args.triangles}-triangle soup Mesh (bvh.rs tree for the reference's visit order; RT_HIT_ORDERED walks a traversal-only tree over the same triangles), "
                               f"{n} incoherent f64 rays per GPU per step", "traversal": args.mode,
                   "bvh_nodes": int(info.n_nodes), "scene_bytes": int(info.device_bytes), "hit_fraction": hit_frac,
                   "scene_build_s": build_s, "scene_describe_s": describe_s, "process_init_s": init_s,
                   "device_ms_per_step": dev_ms},
    }

    if rank == 0 and world_size == 1 and args.mode == "ordered":
        # the same batch with RT_HIT_EXACT_CULL: nearest-first, but every box culled exactly as Aabb::hit culls it
        t_fast = hits["t"].clone()
        ex_flags = F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL
        scene.closest_hit_device(rays, hits, n, seed=0, flags=ex_flags, stream=stream)
        barrier()
        scene.reset_counters()
        t1 = time.perf_counter()
        for _ in range(3):
            scene.closest_hit_device(rays, hits, n, seed=0, flags=ex_flags, stream=stream)
        barrier()
        ex_s = (time.perf_counter() - t1) / 3
        out["exact_cull"] = {"metric": "Mrays/s closest-hit", "flags": "RT_HIT_ORDERED | RT_HIT_EXACT_CULL", "value": n / ex_s / 1e6, "unit": "Mrays/s",
                             "ms_per_step": 1e3 * ex_s, "t_equal_to_conservative_mode": bool(torch.equal(hits["t"], t_fast))}
        scene.closest_hit_device(rays, hits, n, seed=0, flags=flags, stream=stream)  # the parity block below reads the headline mode's results
        torch.cuda.synchronize()
        del t_fast

    if rank == 0:
        import oracle_ffi as O  # the CPU oracle: checker and reported baseline only
        cpu = O.OracleScene(desc)
        out["roofline"] = _roofline(args, n, dev_ms, box_tests, prim_tests, cpu, o, d, tm, np, shader_khz)
        if not args.no_cpu and world_size == 1:
            threads = min(16, os.cpu_count() or 1)
            mc = min(args.cpu_sample, n)
            t1 = time.perf_counter()
            ref = cpu.closest_hit(o[:mc], d[:mc], time=tm[:mc], nthreads=threads)
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": mc / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": f"first {mc} rays of the same batch, C++ f64 restatement of the reference "
                                             f"(reference-order traversal), {threads} threads, {cpu_s:.1f} s"}
            m1 = min(mc, max(1, mc // 16))  # the same restatement on one core, on a smaller slice of the batch
            t1 = time.perf_counter()
            cpu.closest_hit(o[:m1], d[:m1], time=tm[:m1], nthreads=1)
            one_s = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": m1 / one_s / 1e6, "unit": "Mrays/s", "sample": f"first {m1} rays, {one_s:.1f} s"}
            out["config"]["parity_vs_oracle"] = _parity(hits, ref, mc, np)
        del cpu
    scene.close()

    # ------------------------------------------------------------------ the same batch size on a closed mesh of >= 1M triangles
    if rank == 0 and world_size == 1 and not args.no_mesh:
        out["mesh"] = _mesh_figure(rt, F, scenes, np, n, device_rays, timed_steps, actual_counts, hits, args)

    if rank == 0 and world_size == 1 and not args.no_render:
        out["render"] = {name: _render_figure(rt, F, scenes, np, name, with_cpu=not args.no_cpu) for name in ("book1_final", "book2_final", "cornell_box")}

    # ------------------------------------------------------------------ the frame partitioned over the ranks + one reduce
    if dist is not None and not args.no_render:
        try:
            fig = _multi_gpu_render(rt, F, scenes, parallel, torch, dist, dev, stream, args.frame_strata * world_size)
        except Exception as e:  # noqa: BLE001 -- the closest-hit line above has been measured: report it whatever happens to this leg
            fig = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["multi_gpu_render"] = fig
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


VALU_NODE, VALU_PRIM, VALU_RAY = 45, 70, 40  # lane-instructions per box test, per triangle test, per ray
CUS, LANES = 256, 64


def _roofline(args, n, dev_ms, box_tests, prim_tests, cpu, o, d, tm, np, shader_khz):
    """Two roofs for the dominant kernel; `bound` names the one that binds.

    HBM: algorithmic bytes per ray = 64 per node visit + 72 per triangle test + 72 (ray in) + 48 (hit out), with the
    visits THIS kernel made (a visit of a 4-wide node = 4 box tests), times the rays of a launch, over the kernel's
    average duration in the timed region (HIP events of the library on the launch stream), against 8 TB/s.  Beside
    it `frac_requested` (the bytes the kernel's own record formats ask for: 48-byte nodes, 80 bytes of a triangle's
    line, the 64-byte prepared ray, 12 bytes out) and, from the counter passes of the same command kept under
    profiles/, the bytes that really crossed the fabric (`traffic`, `traffic_frac`).
    Vector ALU: lane-instructions per ray = 45 per box test + 70 per triangle test + 40, over the time, against
    256 CUs x 64 lanes x the shader clock measured during a launch of this kernel.
    The same two formulas on the REFERENCE-ORDER counts of the oracle (left-then-right traversal of the same tree) are
    reported as `ref_order`: they say how much work the ordered traversal avoids, not how fast the kernel is.
    Which roof binds: with the counter passes of this build at hand, the share of cycles the vector ALUs are issuing
    (`valu.busy_frac`: idle lanes of an instruction still take the slot) against the share of the HBM peak that really
    crosses the fabric (`traffic_frac`); without them, `valu.frac` against `frac_requested`.  `achieved`, `peak`, `unit`
    and `frac` at the top level are the HBM figures on the bytes of the kernel's own visits in either case."""
    kernel_ms = dev_ms["traversal_ms"]
    ordered = args.mode == "ordered"
    node_visits = box_tests / 4.0 if ordered else box_tests
    own_bytes = NODE_BYTES * node_visits + TRI_BYTES * prim_tests + RAY_BYTES + HIT_BYTES
    per_s = n / (kernel_ms * 1e-3)
    m = min(args.count_sample, n)
    ctr = {}
    cpu.closest_hit(o[:m], d[:m], time=tm[:m], nthreads=min(16, os.cpu_count() or 1), counters=ctr)
    ref_nodes, ref_prims = ctr["node_visits"] / m, ctr["prim_tests"] / m
    ref_bytes = NODE_BYTES * ref_nodes + TRI_BYTES * ref_prims + RAY_BYTES + HIT_BYTES
    hbm = {"achieved": own_bytes * per_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": own_bytes * per_s / 1e9 / HBM_PEAK_GBS,
           "alg_bytes_per_ray": own_bytes}
    if ordered:
        hbm["requested_bytes_per_ray"] = 48 * node_visits + 80 * prim_tests + 64 + 12
        hbm["frac_requested"] = hbm["requested_bytes_per_ray"] * per_s / 1e9 / HBM_PEAK_GBS
    clock_ghz = shader_khz / 1e6 if shader_khz else 2.4
    valu_peak = CUS * LANES * clock_ghz  # G lane-instructions/s
    own_instr = VALU_NODE * box_tests + VALU_PRIM * prim_tests + VALU_RAY
    ref_instr = VALU_NODE * ref_nodes + VALU_PRIM * ref_prims + VALU_RAY
    valu = {"achieved": own_instr * per_s / 1e9, "peak": valu_peak, "unit": "G lane-instr/s", "frac": own_instr * per_s / 1e9 / valu_peak,
            "lane_instr_per_ray": own_instr, "clock_ghz": clock_ghz, "clock_measured": bool(shader_khz)}
    traffic = None
    prof = _profile_for_this_build()
    extra = {}
    if prof:
        # fabric-side bytes per launch from the rocprofv3 --pmc passes of the same command (profiles/), valid for this
        # build only (source hash): L2 <-> Infinity Cache / HBM reads (FETCH_SIZE x 2 on gfx950) + writes
        traffic = prof.get("hbm_bytes_per_launch")
        if traffic:
            hbm["traffic_gbs"] = traffic / (kernel_ms * 1e-3) / 1e9
            hbm["traffic_frac"] = hbm["traffic_gbs"] / HBM_PEAK_GBS
        for k in ("l2_hit_rate", "valu_busy_frac", "valu_lane_utilisation", "wave_wait_frac", "profile"):
            if k in prof:
                extra[k] = prof[k]
    if "valu_busy_frac" in extra and "traffic_frac" in hbm:
        # counters of this build: the share of cycles the vector ALUs issue (whatever the lanes do) against the share of
        # the HBM peak that crosses the fabric
        valu["busy_frac"], valu["lane_utilisation"] = extra["valu_busy_frac"], extra.get("valu_lane_utilisation")
        bound = "valu" if valu["busy_frac"] >= hbm["traffic_frac"] else "hbm"
        basis = (f"counters of this build: vector ALU busy {valu['busy_frac']:.2f} of the cycles at lane utilisation "
                 f"{valu['lane_utilisation'] or 0:.2f}; fabric traffic {hbm['traffic_frac']:.2f} of the HBM peak")
    else:
        hbm_binding = hbm.get("frac_requested", hbm["frac"])
        bound = "valu" if valu["frac"] >= hbm_binding else "hbm"
        basis = "formulas only (no counter profile of this build under profiles/): valu.frac against hbm.frac_requested"
    # achieved / peak / unit / frac stay on the bytes of the kernel's own visits whichever roof binds; the vector-ALU roof is `valu`
    r = {"bound": bound, "bound_basis": basis, "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"], "traffic": traffic,
         "kernel": "bvh4_closest_hit_kernel" if ordered else "closest_hit_kernel", "kernel_ms": kernel_ms,
         "node_visits_per_ray": node_visits, "box_tests_per_ray": box_tests, "prim_tests_per_ray": prim_tests,
         "hbm": hbm, "valu": valu,
         "ref_order": {"nodes_per_ray": ref_nodes, "prims_per_ray": ref_prims, "bytes_per_ray": ref_bytes, "lane_instr_per_ray": ref_instr,
                       "hbm_frac": ref_bytes * per_s / 1e9 / HBM_PEAK_GBS, "valu_frac": ref_instr * per_s / 1e9 / valu_peak,
                       "visit_reduction": ref_bytes / own_bytes, "count_sample_rays": m}}
    r.update(extra)
    return r


def _profile_for_this_build():
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        prof = json.load(open(p))
    except Exception:
        return None
    return prof if prof.get("source_hash") == source_hash() else None


def _parity(hits, ref, mc, np):
    got = hits["prim"][:mc].cpu().numpy().astype(np.uint32)
    same = got == ref["prim"]
    t_gpu = hits["t"][:mc].cpu().numpy()
    tie = np.isclose(t_gpu, ref["t"], rtol=1e-6, atol=0) | (np.isinf(t_gpu) & np.isinf(ref["t"]))
    return {"rays": int(mc), "prim_mismatch_not_tie": int((~same & ~tie).sum()), "prim_mismatch_tie": int((~same & tie).sum()),
            "t_bit_identical": bool(np.array_equal(t_gpu, ref["t"]))}


def _mesh_figure(rt, F, scenes, np, n, device_rays, timed_steps, actual_counts, hits, args):
    """Closest hit on a closed 'bunny-class' surface of 1,310,720 triangles (icosphere, 8 subdivisions, smooth radial
    displacement; no OBJ asset ships with the reference), same batch shape as the headline."""
    import oracle_ffi as O
    tris = scenes.icosphere_mesh(subdiv=8)
    desc = rt.SceneDesc.from_world(rt.Mesh(tris, rt.Lambertian(rt.Color(.5, .5, .5))))
    t0 = time.time()
    scene = rt.Scene(desc)
    build_s = time.time() - t0  # rt_scene_create alone, as for the headline scene
    b = scene.info().world_bounds
    o, d, tm = scenes.incoherent_rays(n, [b[k] for k in range(6)], seed=5)
    rays = device_rays(o, d, tm)
    elapsed, dev_ms = timed_steps(scene, rays, args.steps, args.warmup)
    box_tests, prim_tests, shader_khz = actual_counts(scene, rays)
    fig = {"metric": "Mrays/s closest-hit", "scene": f"closed mesh, {len(tris)} triangles (icosphere subdiv 8, displaced), {n} incoherent f64 rays",
           "value": float(n) * args.steps / elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": 1e3 * elapsed / args.steps,
           "device_ms_per_step": dev_ms, "scene_build_s": build_s, "hit_fraction": float((hits["prim"] != -1).double().mean().item()),
           "node_visits_per_ray": box_tests / 4.0, "prim_tests_per_ray": prim_tests}
    own_bytes = NODE_BYTES * box_tests / 4.0 + TRI_BYTES * prim_tests + RAY_BYTES + HIT_BYTES
    per_s = n / (dev_ms["traversal_ms"] * 1e-3)
    fig["hbm_frac"] = own_bytes * per_s / 1e9 / HBM_PEAK_GBS
    fig["valu_frac"] = (VALU_NODE * box_tests + VALU_PRIM * prim_tests + VALU_RAY) * per_s / 1e9 / (CUS * LANES * (shader_khz / 1e6 if shader_khz else 2.4))
    if not args.no_cpu:
        m = min(1 << 16, n)
        ref = O.OracleScene(desc).closest_hit(o[:m], d[:m], time=tm[:m], nthreads=min(16, os.cpu_count() or 1))
        fig["parity_vs_oracle"] = _parity(hits, ref, m, np)
    scene.close()
    return fig


def _render_figure(rt, F, scenes, np, name, with_cpu):
    """Second half of the metric: camera samples/s of a full frame at the configuration BASELINE.json names, next to the
    oracle on some rows of it.
    book1_final: the Book-1 final scene (main.rs:99-202), 1200x675, 10 samples per pixel asked (sqrt_spp = 3: 9 taken,
                 camera.rs:170-171), depth 50, defocus blur, no light list (material pdf alone);
    book2_final: final_scene (main.rs:651-788), 800x800, depth 40 (BVHs, instances, media, textures), 100 samples per
                 pixel (the reference's own run used 10000, which only lengthens the run);
    cornell_box: the scene live at the reference's HEAD (main.rs:421-551), 600x600, 1000 asked (961 taken), depth 50,
                 mixture PDF."""
    if name == "book1_final":
        world, lights, cam = scenes.book1_final(seed=0, image_width=1200, samples_per_pixel=10, max_depth=50)
    elif name == "book2_final":
        world, lights, cam = scenes.book2_final(image_width=800, samples_per_pixel=100, max_depth=40)
    else:
        world, lights, cam = scenes.cornell_box(image_width=600, samples_per_pixel=1000, max_depth=50)
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    c, d = cam.to_struct(), cam.derived()
    n_strata = d.sqrt_spp * d.sqrt_spp
    paths = cam.image_width * d.image_height * n_strata
    gpu.render(c, seed=1, work=F.rt_work(0, 0, cam.image_width, 8, 0, 1))  # workspace + code
    t0 = time.perf_counter()
    accum = gpu.render(c, seed=1)
    wall = time.perf_counter() - t0
    cnt = gpu.counters()
    fig = {"metric": "camera samples/s", "scene": f"{name} {cam.image_width}x{d.image_height}, {n_strata} samples/pixel, depth {cam.max_depth}",
           "value": paths / wall, "unit": "samples/s", "ms_frame": 1e3 * wall, "device_ms": cnt.ms_last_render,
           "extend_queries": int(cnt.render_rays)}
    # Bytes of path state the wavefront kernels move per path-bounce (a world.hit query and the shading of its result):
    # extend reads the 128-byte path record and writes its 16-byte hit block, shade reads the record and writes it back,
    # and the path's 4-byte queue entry is read by both and written once; per path, raygen writes the record once and
    # accumulate reads the 24 bytes of radiance.  Scene data (nodes, primitives, materials) is not counted: it is shared
    # by the paths in flight and mostly served on-die.
    bounces = max(1, int(cnt.render_rays))
    per_bounce = (128 + 16) + (128 + 128) + 3 * 4 + (128 + 24) * paths / bounces
    fig["roofline"] = {"bound": "hbm", "bytes_per_path_bounce": per_bounce, "achieved": per_bounce * bounces / (cnt.ms_last_render * 1e-3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_bounce * bounces / (cnt.ms_last_render * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if with_cpu:
        import oracle_ffi as O
        # rows of the frame the oracle renders beside it: a bounded sample (a few seconds of the host's cores)
        rows, threads = {"book1_final": 675, "book2_final": 200, "cornell_box": 48}[name], min(16, os.cpu_count() or 1)
        rows = min(rows, d.image_height)
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
    gpu.close()
    return fig


def _multi_gpu_render(rt, F, scenes, parallel, torch, dist, dev, stream, strata_total):
    """The 4096 x 2304 mesh + spheres frame (4096 strata) as an N-rank job: every rank builds the scene on ITS device,
    renders its band of the first `strata_total` strata with rt_render_device, then the frame-end exchange of the f64
    accumulator over RCCL -- measured twice, once per exchange: one reduce (sum) to rank 0, and a reduce-scatter by
    band owner followed by a gather of the bands to rank 0 (ray_tracer_amd.parallel.collect_accumulator).  The frame
    does not depend on the partition (streams are keyed by pixel and stratum), so both jobs must produce the same
    sums; rank 0 checks that.  With one rank (--force-dist) the collectives still run on the device accumulator."""
    world, lights, cam = scenes.mesh_and_spheres(image_width=4096, samples_per_pixel=4096, max_depth=50)
    scene = rt.Scene(rt.SceneDesc.from_world(world, lights))  # created with this rank's device current
    c, d = cam.to_struct(), cam.derived()
    n_strata = d.sqrt_spp * d.sqrt_spp

    def render_unit(work, acc):
        scene.render_device(c, acc, seed=7, work=work, stream=stream)

    # warm-up on one stratum per rank (path workspace, code objects, RCCL channel set-up of every collective used)
    for coll in ("reduce", "reduce_scatter"):
        warm = torch.zeros((d.image_height, cam.image_width, 3), dtype=torch.float64, device=dev)
        parallel.render_frame_job(render_unit, warm, cam.image_width, d.image_height, n_strata, dist=dist, strata_limit=dist.get_world_size(),
                                  sync=torch.cuda.synchronize, collective=coll)
    figs, sums = {}, {}
    for coll in ("reduce", "reduce_scatter"):
        accum = torch.zeros((d.image_height, cam.image_width, 3), dtype=torch.float64, device=dev)
        dist.barrier()
        accum, tm = parallel.render_frame_job(render_unit, accum, cam.image_width, d.image_height, n_strata, dist=dist,
                                              strata_limit=strata_total, sync=torch.cuda.synchronize, collective=coll)
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, tm)
        total_s = max(g["render_s"] + g["reduce_s"] for g in gathered)
        paths = sum(g["paths"] for g in gathered)
        figs[coll] = {"value": paths / total_s, "unit": "samples/s", "paths": paths, "per_rank_render_ms": [1e3 * g["render_s"] for g in gathered],
                      "per_rank_exchange_ms": [1e3 * g["reduce_s"] for g in gathered]}
        if dist.get_rank() == 0:
            sums[coll] = float(torch.nan_to_num(accum).sum().item())
            figs[coll]["mean_radiance"] = float(torch.nan_to_num(accum).mean().item() / max(1, strata_total))
    best = max(figs, key=lambda k: figs[k]["value"])
    fig = {"metric": "camera samples/s", "scene": f"mesh + spheres {cam.image_width}x{d.image_height}, strata [0, {strata_total}) of {n_strata}, depth {cam.max_depth}",
           "partition": "strata", "nccl_world_size": dist.get_world_size(), "backend": dist.get_backend(), "value": figs[best]["value"], "unit": "samples/s",
           "exchange_of_value": best, "accumulator_bytes": int(d.image_height) * int(cam.image_width) * 3 * 8,
           "exchanges": {"reduce": dict(figs["reduce"], collective="reduce(sum) of the f64 accumulator to rank 0 (nccl = RCCL)"),
                         "reduce_scatter": dict(figs["reduce_scatter"], collective="reduce_scatter(sum) by band owner + gather of the bands to rank 0 (nccl = RCCL)")}}
    if dist.get_rank() == 0:
        fig["exchanges_agree"] = bool(abs(sums["reduce"] - sums["reduce_scatter"]) <= 1e-9 * max(1.0, abs(sums["reduce"])))
    scene.close()
    return fig


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    main()
