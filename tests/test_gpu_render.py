"""GPU parity of the wavefront path tracer (raygen / extend / shade / compact / accumulate
kernels behind rt_render) against the oracle's recursive Camera::render with the same
counter-based RNG streams.  Bar: per-pixel RMSE of the scaled linear radiance <= 1e-3
(the recursion and the iterative throughput form round differently, and device
transcendentals differ from glibc's in the last ulps)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F
from ray_tracer_amd import scenes

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-3


def scaled(cam, accum):
    d = O.camera_initialize(cam)
    return np.nan_to_num(accum * d.pixel_samples_scale, nan=0.0, posinf=1e6, neginf=0.0)


def render_both(build, width, spp, depth, seed=1, **kw):
    world, lights, cam = build(**kw)
    cam.image_width, cam.samples_per_pixel, cam.max_depth = width, spp, depth
    desc = rt.SceneDesc.from_world(world, lights)
    gpu, cpu = rt.Scene(desc), O.OracleScene(desc)
    c = cam.to_struct()
    stats = {}
    a = gpu.render(c, seed=seed)
    b = cpu.render(c, seed=seed, stats=stats)
    ctr = gpu.counters()
    return cam, c, a, b, stats, ctr, gpu


def check(c, a, b, tol=RMSE_TOL):
    sa, sb = scaled(c, a), scaled(c, b)
    rmse = float(np.sqrt(np.mean((sa - sb) ** 2)))
    # most pixels agree to rounding; only paths that cross a decision boundary may differ
    close = np.isclose(sa, sb, rtol=1e-9, atol=1e-12).mean()
    assert rmse <= tol, (rmse, close)
    assert close >= 0.999, close  # measured on these frames: every pixel within 1e-9 relative, RMSE 1e-17 .. 1e-13
    assert sb.mean() > 1e-3, "oracle image is black: the test exercises nothing"
    return rmse


def assert_same_ray_count(gpu_rays, oracle_rays):
    """Both sides issue one world.hit per path and bounce.  The device's sin/cos of the azimuth samples and
    glibc's may differ in the last place, which now and then flips a decision along a path (a grazing hit, a
    Schlick draw) and with it the path's length: the totals agree to a few in 10^5, not exactly."""
    assert abs(int(gpu_rays) - int(oracle_rays)) <= max(2, int(1e-4 * oracle_rays)), (gpu_rays, oracle_rays)


def test_russian_roulette_switch():
    """rt_camera.russian_roulette = 1: the termination rule of the reference's history (camera.rs:407-416), in the
    oracle and on the device alike -- same frames, same number of world.hit queries, fewer of them than without
    the rule (the reference's own timings: 11756 s -> 2793 s for the Book-2 final scene, time_record.txt:22-23)."""
    for build, width, spp, depth in ((scenes.cornell_box, 64, 64, 50), (scenes.book2_final, 48, 16, 40), (scenes.cornell_smoke, 48, 36, 50),
                                     (scenes.book1_final, 64, 9, 50)):
        world, lights, cam = build()
        cam.image_width, cam.samples_per_pixel, cam.max_depth = width, spp, depth
        desc = rt.SceneDesc.from_world(world, lights)
        gpu, cpu = rt.Scene(desc), O.OracleScene(desc)
        plain = gpu.render(cam.to_struct(), seed=2)
        rays_plain = gpu.counters().render_rays
        cam.russian_roulette = True
        c = cam.to_struct()
        assert c.russian_roulette == 1
        stats = {}
        gpu.reset_counters()
        a, b = gpu.render(c, seed=2), cpu.render(c, seed=2, stats=stats)
        check(c, a, b)
        assert_same_ray_count(gpu.counters().render_rays, stats["rays"])
        assert gpu.counters().render_rays < 0.9 * rays_plain, (build.__name__, gpu.counters().render_rays, rays_plain)
        assert not np.array_equal(np.nan_to_num(a), np.nan_to_num(plain))
        gpu.close()


def test_cornell_box_mixture_pdf():
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.cornell_box, 64, 64, 50)
    check(c, a, b)
    # every top-level world.hit of the oracle corresponds to one extend query on the GPU
    assert_same_ray_count(ctr.render_rays, stats["rays"])
    assert ctr.paths == 64 * 64 * 64
    img = rt.Image(cam, a)
    ref = O.resolve_rgb8(c, b)
    assert np.mean(np.abs(img.rgb.astype(int) - ref.astype(int)) <= 1) > 0.995


def test_book1_final_no_lights_defocus():
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.book1_final, 96, 9, 50, seed=2)
    check(c, a, b)


def test_book1_moving_checker():
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.book1_final, 80, 4, 20, seed=3, moving=True, checker_ground=True)
    check(c, a, b)


def test_book2_final_media_textures_instances():
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.book2_final, 64, 16, 40, seed=4)
    check(c, a, b)
    assert_same_ray_count(ctr.render_rays, stats["rays"])


def test_cornell_smoke():
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.cornell_smoke, 48, 16, 50, seed=5)
    check(c, a, b)


def _nested_fog():
    """Cornell smoke with one block's fog held inside a larger, thinner fog (a medium within a medium's boundary)."""
    world, lights, cam = scenes.cornell_smoke()
    glass = rt.Dielectric(1.5)
    inner = rt.ConstantMedium.new_with_color(rt.Sphere(rt.Point3(278.0, 278.0, 278.0), 90.0, glass), 0.02, rt.Color(.9, .9, .9))
    shell = rt.HittableList()
    shell.add(inner)
    shell.add(rt.Sphere(rt.Point3(278.0, 278.0, 278.0), 150.0, glass))
    world.add(rt.ConstantMedium.new_with_color(shell, 0.004, rt.Color(.3, .5, .9)))
    return world, lights, cam


def test_nested_media_render():
    cam, c, a, b, stats, ctr, gpu = render_both(_nested_fog, 40, 9, 30, seed=11)
    check(c, a, b)
    assert_same_ray_count(ctr.render_rays, stats["rays"])


def _image_fog():
    """Cornell smoke whose fogs scatter with an image texture: the texture is looked up at the u,v the hit record held
    before the medium was hit (walls and box sides: quads write u,v; the default record: 0, 0)."""
    world, lights, cam = scenes.cornell_smoke()
    img = rt.ImageTexture(scenes.procedural_image())
    world.add(rt.ConstantMedium.new_with_texture(rt.Sphere(rt.Point3(278.0, 400.0, 278.0), 100.0, rt.Dielectric(1.5)), 0.02, img))
    bag = rt.HittableList()
    bag.add(rt.Sphere(rt.Point3(150.0, 100.0, 150.0), 40.0, rt.Lambertian(rt.Color(.8, .3, .3))))
    bag.add(rt.ConstantMedium.new_with_texture(rt.Sphere(rt.Point3(150.0, 100.0, 150.0), 90.0, rt.Dielectric(1.5)), 0.03, img))
    world.add(bag)
    return world, lights, cam


def test_phase_function_over_an_image_texture():
    for depth in (30, 6):  # 6: the tail kernel takes over early on so few paths
        cam, c, a, b, stats, ctr, gpu = render_both(_image_fog, 40, 9, depth, seed=13)
        check(c, a, b)
        assert_same_ray_count(ctr.render_rays, stats["rays"])


def _two_quad_lights():
    """The Cornell box with a second quad light on the left wall, named in the light list between the ceiling light
    and the glass sphere: lanes of one wave sample different quads."""
    world, lights, cam = scenes.cornell_box()
    world.add(rt.Quad(rt.Point3(554.0, 200.0, 200.0), rt.Vec3(0.0, 0.0, 150.0), rt.Vec3(0.0, 150.0, 0.0), rt.DiffuseLight.from_color(rt.Color(4.0, 6.0, 9.0))))
    members = list(lights.objects)
    lights = rt.HittableList()
    lights.add(members[0])
    lights.add(rt.Quad(rt.Point3(554.0, 200.0, 200.0), rt.Vec3(0.0, 0.0, 150.0), rt.Vec3(0.0, 150.0, 0.0), rt.Lambertian(rt.Color(0, 0, 0))))
    for m in members[1:]:
        lights.add(m)
    return world, lights, cam


def test_light_list_with_two_quads_and_a_sphere():
    cam, c, a, b, stats, ctr, gpu = render_both(_two_quad_lights, 48, 16, 30, seed=14)
    check(c, a, b)
    assert_same_ray_count(ctr.render_rays, stats["rays"])


def test_texture_scenes():
    for build in (scenes.checkered_spheres, scenes.earth, scenes.perlin_spheres, scenes.simple_light, scenes.quads,
                  scenes.for_output13, scenes.test_triangle):
        cam, c, a, b, stats, ctr, gpu = render_both(build, 48, 9, 50, seed=6)
        check(c, a, b)


def test_earth_from_an_image_file(tmp_path, monkeypatch):
    """texture.rs:81-108 over a decoded file: found through $RTW_IMAGES, shaded on the device as in the oracle;
    a name that is not found shades cyan (texture.rs:92) on both sides."""
    from PIL import Image
    yy, xx = np.mgrid[0:64, 0:128]
    px = np.stack([(xx * 2) % 256, (yy * 4) % 256, (xx + yy) % 256], axis=2).astype(np.uint8)
    Image.fromarray(px, "RGB").save(tmp_path / "map.png")
    monkeypatch.setenv("RTW_IMAGES", str(tmp_path))
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.earth, 64, 9, 50, seed=6, image_file="map.png")
    check(c, a, b)
    cam, c, a2, b2, stats, ctr, gpu = render_both(scenes.earth, 64, 9, 50, seed=6, image_file="absent.png")
    check(c, a2, b2)
    assert not np.array_equal(a, a2)
    centre = scaled(c, a2)[a2.shape[0] // 2, a2.shape[1] // 2]
    assert centre[0] < 0.35 * centre[1] and abs(centre[1] - centre[2]) < 0.35 * centre[1]   # cyan globe: little red


def test_mesh_and_spheres_frame_and_its_partition():
    """The multi-GPU frame (sphere BVH + triangle-mesh BVH in one list, metal mesh): parity with the oracle, and
    the per-rank (tile, stratum-range) units of an 8-way partition add up to the unsplit frame."""
    from ray_tracer_amd import parallel
    cam, c, a, b, stats, ctr, gpu = render_both(scenes.mesh_and_spheres, 96, 16, 12, seed=8, subdiv=3)
    check(c, a, b)
    assert_same_ray_count(ctr.render_rays, stats["rays"])
    d = O.camera_initialize(c)
    acc = np.zeros_like(a)
    for rank in range(8):
        for w in parallel.partition_frame(c.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, rank, 8):
            gpu.render(c, seed=8, work=w, accum=acc)
    assert np.allclose(acc, a, rtol=1e-12, atol=1e-12, equal_nan=True)


def test_partition_independence_and_accumulation():
    """The frame split by tile and by stratum range must reproduce the unsplit frame (the streams are
    keyed by pixel and stratum); batches smaller than the frame must not change it either."""
    world, lights, cam = scenes.cornell_box(image_width=40, samples_per_pixel=16, max_depth=12)
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    c = cam.to_struct()
    full = gpu.render(c, seed=9)
    acc = np.zeros_like(full)
    for (x0, y0, x1, y1) in ((0, 0, 17, 40), (17, 0, 40, 23), (17, 23, 40, 40)):
        for (s0, s1) in ((0, 5), (5, 16)):
            gpu.render(c, seed=9, work=F.rt_work(x0, y0, x1, y1, s0, s1), accum=acc)
    assert np.allclose(acc, full, rtol=1e-12, atol=1e-12, equal_nan=True)
    # strata summed in order within one call reproduce the unsplit sum bit for bit when the tile is whole
    acc2 = np.zeros_like(full)
    gpu.render(c, seed=9, work=F.rt_work(0, 0, 40, 40, 0, 16), accum=acc2)
    assert np.array_equal(np.nan_to_num(acc2), np.nan_to_num(full))
    with pytest.raises(rt.RtError):
        gpu.render(c, seed=9, work=F.rt_work(0, 0, 41, 40, 0, 16))
    with pytest.raises(rt.RtError):
        gpu.render(c, seed=9, work=F.rt_work(0, 0, 40, 40, 0, 17))


def test_small_workspace_batches(monkeypatch):
    import os
    world, lights, cam = scenes.cornell_box(image_width=40, samples_per_pixel=16, max_depth=12)
    desc = rt.SceneDesc.from_world(world, lights)
    c = cam.to_struct()
    full = rt.Scene(desc).render(c, seed=9)
    os.environ["MI355RT_MAX_PATHS"] = "2048"  # forces row tiles and one-stratum batches
    try:
        small = rt.Scene(desc).render(c, seed=9)
    finally:
        del os.environ["MI355RT_MAX_PATHS"]
    assert np.allclose(small, full, rtol=1e-12, atol=1e-12, equal_nan=True)


def test_render_device_entry_point_with_rank_units_on_a_stream():
    """rt_render_device (device accumulator, caller's HIP stream): what every rank of a multi-GPU job calls.
    The units of a 4-rank stratum split and of a 3-rank tile split, rendered one after the other into one
    device accumulator on a side stream, must give the frame rt_render returns."""
    import torch
    from ray_tracer_amd import parallel
    world, lights, cam = scenes.cornell_box(image_width=48, samples_per_pixel=16, max_depth=10)
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    assert torch.cuda.is_available()
    c = cam.to_struct()
    d = O.camera_initialize(c)
    full = gpu.render(c, seed=11)
    stream = torch.cuda.Stream()
    for mode, ranks in (("strata", 4), ("tiles", 3)):
        acc = torch.zeros((d.image_height, c.image_width, 3), dtype=torch.float64, device="cuda")
        with torch.cuda.stream(stream):
            for rank in range(ranks):
                for w in parallel.partition_frame(c.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, rank, ranks, mode=mode):
                    gpu.render_device(c, acc, seed=11, work=w, stream=stream.cuda_stream)
        stream.synchronize()
        assert np.allclose(acc.cpu().numpy(), full, rtol=1e-12, atol=1e-12, equal_nan=True), mode
    # the single-process form of the distributed driver (no process group): same frame
    acc = torch.zeros((d.image_height, c.image_width, 3), dtype=torch.float64, device="cuda")
    parallel.render_distributed(lambda w, a: gpu.render_device(c, a, seed=11, work=w, stream=torch.cuda.current_stream().cuda_stream),
                                acc, c.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), full, rtol=1e-12, atol=1e-12, equal_nan=True)


def test_frame_job_exchanges_run_on_the_device():
    """render_frame_job as bench.py --gpus N runs it, here with a process group of one rank on RCCL: every pairing of
    partition and frame-end exchange (strata + reduce, strata + reduce_scatter by band owner + gather, tiles + gather,
    tiles + reduce) goes through the collective on the device accumulator and must return the unsplit frame."""
    import socket
    import torch
    import torch.distributed as dist
    from ray_tracer_amd import parallel
    world, lights, cam = scenes.cornell_box(image_width=48, samples_per_pixel=16, max_depth=10)
    gpu = rt.Scene(rt.SceneDesc.from_world(world, lights))
    c = cam.to_struct()
    d = O.camera_initialize(c)
    full = gpu.render(c, seed=11)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        unit = lambda w, a: gpu.render_device(c, a, seed=11, work=w, stream=torch.cuda.current_stream().cuda_stream)
        for mode, collective in (("strata", "reduce"), ("strata", "reduce_scatter"), ("tiles", "gather"), ("tiles", "reduce")):
            acc = torch.zeros((d.image_height, c.image_width, 3), dtype=torch.float64, device="cuda")
            acc, tm = parallel.render_frame_job(unit, acc, c.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, dist=dist, mode=mode,
                                                sync=torch.cuda.synchronize, collective=collective)
            assert tm["collective"] == collective and tm["reduce_s"] > 0.0
            assert tm["paths"] == c.image_width * d.image_height * d.sqrt_spp * d.sqrt_spp
            assert np.allclose(acc.cpu().numpy(), full, rtol=1e-12, atol=1e-12, equal_nan=True), (mode, collective)
    finally:
        dist.destroy_process_group()


def test_shading_variants_agree_bit_for_bit(monkeypatch):
    """Scenes whose textures are all solid colours are shaded by a kernel compiled without the texture code
    (fewer registers, one more wave per SIMD).  MI355RT_SHADE_TEXTURES=1 forces the full kernel: two
    independently compiled kernels must produce the same accumulator, bit for bit."""
    cases = [scenes.cornell_box(image_width=120, samples_per_pixel=16, max_depth=50),
             scenes.cornell_smoke(image_width=96, samples_per_pixel=9, max_depth=50),
             scenes.mesh_and_spheres(image_width=96, samples_per_pixel=4, max_depth=20)]
    w1, l1, c1 = scenes.book1_final(seed=2)  # metal, glass and diffuse spheres, defocus blur, no light list
    c1.image_width, c1.samples_per_pixel, c1.max_depth = 120, 9, 30
    cases.append((w1, l1, c1))
    for world, lights, cam in cases:
        g = rt.Scene(rt.SceneDesc.from_world(world, lights))
        c = cam.to_struct()
        monkeypatch.delenv("MI355RT_SHADE_TEXTURES", raising=False)
        plain = g.render(c, seed=3).copy()
        monkeypatch.setenv("MI355RT_SHADE_TEXTURES", "1")
        full = g.render(c, seed=3).copy()
        assert np.array_equal(plain, full, equal_nan=True)


def test_tail_kernel_agrees_bit_for_bit(monkeypatch):
    """The last bounces of a batch run in one launch in which every lane follows its path to the end
    (MI355RT_TAIL_PATHS: the live-path count at which it takes over; 0 = never).  Whatever the switch-over
    point, the accumulator must be the same, bit for bit, and the extend-query count too."""
    cases = [scenes.cornell_box(image_width=96, samples_per_pixel=16, max_depth=50),
             scenes.cornell_smoke(image_width=64, samples_per_pixel=9, max_depth=50),
             scenes.book2_final(seed=0, image_width=64, samples_per_pixel=4, max_depth=40),
             scenes.mesh_and_spheres(image_width=96, samples_per_pixel=4, max_depth=20)]
    for world, lights, cam in cases:
        g = rt.Scene(rt.SceneDesc.from_world(world, lights))
        c = cam.to_struct()
        out = {}
        for tail in ("0", "1000", "1073741824"):  # never / late / from the second bounce on
            monkeypatch.setenv("MI355RT_TAIL_PATHS", tail)
            g.reset_counters()
            out[tail] = (g.render(c, seed=5).copy(), g.counters().render_rays)
        for tail in ("1000", "1073741824"):
            assert np.array_equal(out["0"][0], out[tail][0], equal_nan=True), tail
            assert out["0"][1] == out[tail][1], tail


def test_nan_samples_land_on_the_same_pixels():
    """A light-sampling pdf of 0/0 makes the reference's sample NaN even when everything after it is black:
    the recursion multiplies the black radiance into the NaN.  Here a diffuse ball sits inside a spherical
    light (sphere.rs:117-131 takes the square root of a negative number for origins inside the sphere) and
    every scattered ray ends on the light's back face, which emits black (material.rs:237-256).  The
    throughput form must carry that black through its product: same NaN pixels as the oracle."""
    from ray_tracer_amd.scenes import (Camera, Color, DiffuseLight, HittableList, Lambertian, Point3, Sphere, Vec3)
    world = HittableList()
    light = Sphere(Point3(0.0, 0.0, 0.0), 10.0, DiffuseLight.from_color(Color(4.0, 4.0, 4.0)))
    world.add(light)
    world.add(Sphere(Point3(0.0, 0.0, 0.0), 1.0, Lambertian(Color(0.7, 0.6, 0.5))))
    lights = HittableList()
    lights.add(light)
    cam = Camera()
    cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 1.0, 32, 4, 6
    cam.background = Color(0.3, 0.4, 0.5)
    cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 40.0, Point3(0.0, 0.0, -5.0), Point3(0.0, 0.0, 0.0), Vec3(0.0, 1.0, 0.0)
    cam.defocus_angle = 0.0
    desc = rt.SceneDesc.from_world(world, lights)
    c = cam.to_struct()
    a = rt.Scene(desc).render(c, seed=12)
    b = O.OracleScene(desc).render(c, seed=12)
    assert np.isnan(b).any() and not np.isnan(b).all()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True)


def test_hoisted_traversal_agrees_with_the_general_kernel(monkeypatch):
    """Worlds (or first world-list members) without instances and media are traced by the primitive-BVH kernel
    over a BVH of their primitives; MI355RT_FAST_EXTEND=0 keeps the general traversal of the scene graph as it
    was declared.  The closest hit does not depend on the grouping, so the frames agree (exact ties between
    two primitives aside, which may pick the other material)."""
    cases = [scenes.book2_final(seed=0, image_width=64, samples_per_pixel=4, max_depth=40),  # first member hoisted
             scenes.mesh_and_spheres(image_width=96, samples_per_pixel=4, max_depth=20)]     # whole world hoisted
    w1, l1, c1 = scenes.book1_final(seed=2)                                                   # one BVH: used as it is
    c1.image_width, c1.samples_per_pixel, c1.max_depth = 96, 4, 30
    cases.append((w1, l1, c1))
    for world, lights, cam in cases:
        g = rt.Scene(rt.SceneDesc.from_world(world, lights))
        c = cam.to_struct()
        monkeypatch.delenv("MI355RT_FAST_EXTEND", raising=False)
        fast = g.render(c, seed=7).copy()
        monkeypatch.setenv("MI355RT_FAST_EXTEND", "0")
        general = g.render(c, seed=7).copy()
        same = np.isclose(fast, general, rtol=1e-12, atol=1e-15, equal_nan=True).all(axis=2).mean()
        assert same >= 0.999, same


def test_render_edge_cases():
    """Frames at the edges of the parameter space: an empty world (every ray is background, camera.rs:258-260),
    a world that is one bare primitive (no list around it), depth 0 (ray_color returns black before it looks at
    the world, camera.rs:252-254), a 1x1 frame with one sample, and a samples_per_pixel that is not a square
    (the stratification truncates it, camera.rs:139-141).  Same accumulator as the oracle in each."""
    from ray_tracer_amd.scenes import Camera, Color, HittableList, Lambertian, Metal, Point3, Sphere, Vec3

    def camera(width, spp, depth):
        cam = Camera()
        cam.aspect_ratio, cam.image_width, cam.samples_per_pixel, cam.max_depth = 1.0, width, spp, depth
        cam.background = Color(0.3, 0.4, 0.5)
        cam.vfov, cam.lookfrom, cam.lookat, cam.vup = 40.0, Point3(0.0, 0.0, -5.0), Point3(0.0, 0.0, 0.0), Vec3(0.0, 1.0, 0.0)
        cam.defocus_angle = 0.0
        return cam

    ball = lambda: Sphere(Point3(0.0, 0.0, 0.0), 1.0, Lambertian(Color(0.5, 0.5, 0.5)))
    mirror = Sphere(Point3(0.0, 0.0, 0.0), 1.0, Metal(Color(0.8, 0.7, 0.6), 0.1))
    cases = [("empty world", HittableList(), camera(8, 4, 5)), ("bare primitive", ball(), camera(8, 4, 5)),
             ("bare metal primitive", mirror, camera(8, 4, 5)), ("depth 0", ball(), camera(8, 4, 0)),
             ("depth 1", ball(), camera(8, 4, 1)), ("1x1 frame, one sample", ball(), camera(1, 1, 5)),
             ("3 samples asked, 1 stratum", ball(), camera(5, 3, 5)), ("8 samples asked, 4 strata", ball(), camera(5, 8, 5))]
    for name, world, cam in cases:
        desc = rt.SceneDesc.from_world(world, None)
        c = cam.to_struct()
        gpu = rt.Scene(desc)
        a = gpu.render(c, seed=13)
        b = O.OracleScene(desc).render(c, seed=13)
        assert a.shape == b.shape, name
        assert np.allclose(a, b, rtol=1e-12, atol=1e-15, equal_nan=True), name
        d = O.camera_initialize(c)
        assert gpu.counters().paths == c.image_width * d.image_height * d.sqrt_spp * d.sqrt_spp, name
