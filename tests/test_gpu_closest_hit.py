"""GPU parity: rt_closest_hit (HIP kernels, through the C ABI) against the CPU oracle on
identical seeded ray batches.  Bar: primitive id exact, t / p / normal bit-exact (the
intersection arithmetic is +,-,*,/,sqrt with contraction off on both sides); sphere u,v go
through acos/atan2, whose device and glibc implementations may differ in the last ulps,
so they are held to 1e-12."""
import numpy as np
import pytest

import oracle_ffi as O
import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F
from ray_tracer_amd import scenes

pytestmark = pytest.mark.gpu

EXACT = ("t", "prim", "front_face", "mat", "nx", "ny", "nz", "px", "py", "pz")


def compare(desc, o, d, time=None, tmin=None, tmax=None, seed=0, min_hits=1, flags=0, medium_uv=False, max_ties=0.02):
    gpu = rt.Scene(desc)
    cpu = O.OracleScene(desc)
    a = gpu.closest_hit(o, d, time=time, tmin=tmin, tmax=tmax, seed=seed, flags=flags)
    b = cpu.closest_hit(o, d, time=time, tmin=tmin, tmax=tmax, seed=seed)
    nhit = int((b["prim"] != F.RT_REF_NONE).sum())
    assert nhit >= min_hits, f"only {nhit} oracle hits: the batch does not exercise the scene"
    # A constant-medium hit goes through ln(U) (constant_medium.rs:90): the device and glibc logarithms
    # may differ in the last ulps, so its t and p are held to 1e-12 relative instead of bit equality.
    medium = (b["prim"] >> 28) == F.RT_REF_MEDIUM
    if flags & F.RT_HIT_ORDERED:
        # near-first traversal: the closest t is the same bit for bit; where several primitives are hit
        # at exactly that t another one of them may be reported (bar: ties within 1e-6 relative)
        tie = (a["prim"] != b["prim"])
        assert np.array_equal(a["t"][tie & ~medium], b["t"][tie & ~medium])
        assert tie.mean() < max_ties, tie.mean()
        for f in a:
            a[f] = a[f][~tie]; b[f] = b[f][~tie]
        medium = medium[~tie]
    assert np.array_equal(a["prim"], b["prim"])
    for f in EXACT:
        same = (a[f] == b[f]) | ((a[f] != a[f]) & (b[f] != b[f]))
        bad = np.nonzero(~same & ~medium)[0]
        assert bad.size == 0, (f, bad[:5], a[f][bad[:5]], b[f][bad[:5]], [hex(int(x)) for x in b["prim"][bad[:5]]])
        if a[f].dtype == np.float64:
            assert np.allclose(a[f][medium], b[f][medium], rtol=1e-12, atol=1e-9), f
        else:
            assert np.array_equal(a[f][medium], b[f][medium]), f
    for f in ("u", "v"):
        # u,v of a medium hit are whatever the record held (constant_medium.rs:96-101); the library reproduces them
        # only in scenes where a phase function reads them (medium_uv), elsewhere it reports 0
        sel = np.ones_like(medium) if medium_uv else ~medium
        assert np.allclose(a[f][sel], b[f][sel], rtol=0, atol=1e-12, equal_nan=True), f
    gpu.close()
    return a, b


def rays_for(bounds, n, seed, spread=0.25):
    b = np.array(bounds)
    ext = np.array([b[1] - b[0], b[3] - b[2], b[5] - b[4]])
    big = [b[0] - spread * ext[0], b[1] + spread * ext[0], b[2] - spread * ext[1], b[3] + spread * ext[1],
           b[4] - spread * ext[2], b[5] + spread * ext[2]]
    return scenes.incoherent_rays(n, big, seed=seed)


def test_kat_vectors_on_gpu():
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    r = kat["ray"]
    o, d = np.array([r["o"]], float), np.array([r["d"]], float)
    k = kat["sphere"]
    h = rt.Scene(rt.HittableList.with_object(rt.Sphere(rt.Point3(*k["center"]), k["radius"], m))).closest_hit(o, d)
    assert h["t"][0] == k["t"] and [h["nx"][0], h["ny"][0], h["nz"][0]] == k["normal"] and h["front_face"][0] == 1
    assert abs(h["u"][0] - k["u"]) < 1e-15 and abs(h["v"][0] - k["v"]) < 1e-15
    k = kat["quad"]
    h = rt.Scene(rt.HittableList.with_object(rt.Quad(rt.Point3(*k["q"]), rt.Vec3(*k["u"]), rt.Vec3(*k["v"]), m))).closest_hit(o, d)
    assert h["t"][0] == k["t"] and h["u"][0] == k["alpha"] and h["v"][0] == k["beta"] and h["nz"][0] == 1.0
    k = kat["triangle"]
    h = rt.Scene(rt.HittableList.with_object(rt.Triangle(rt.Point3(*k["v0"]), rt.Point3(*k["v1"]), rt.Point3(*k["v2"]), m))).closest_hit(o, d)
    assert h["t"][0] == k["t"] and h["u"][0] == k["u"] and h["v"][0] == k["v"] and h["nz"][0] == 1.0


def test_mixed_primitives_list_and_bvh():
    g = np.random.default_rng(5)
    m = [rt.Lambertian(rt.Color(.5, .5, .5)), rt.Metal(rt.Color(.8, .8, .8), 0.1), rt.Dielectric(1.5)]
    lst = rt.HittableList()
    for i in range(200):
        c = g.uniform(-5, 5, 3)
        if i % 4 == 0:
            lst.add(rt.Sphere.new_moving(rt.Point3(*c), rt.Point3(*(c + g.uniform(-.5, .5, 3))), g.uniform(0.1, 0.6), m[i % 3]))
        else:
            lst.add(rt.Sphere(rt.Point3(*c), g.uniform(0.1, 0.6), m[i % 3]))
    for i in range(80):
        lst.add(rt.Quad(rt.Point3(*g.uniform(-5, 5, 3)), rt.Vec3(*g.uniform(-1, 1, 3)), rt.Vec3(*g.uniform(-1, 1, 3)), m[i % 3]))
    for i in range(80):
        p = g.uniform(-5, 5, 3)
        lst.add(rt.Triangle(rt.Point3(*p), rt.Point3(*(p + g.uniform(-1, 1, 3))), rt.Point3(*(p + g.uniform(-1, 1, 3))), m[i % 3]))
    o, d, tm = scenes.incoherent_rays(20000, [-6, 6, -6, 6, -6, 6], seed=6)
    compare(rt.SceneDesc.from_world(lst), o, d, time=tm, min_hits=3000)
    compare(rt.SceneDesc.from_world(rt.BvhNode(lst)), o, d, time=tm, min_hits=3000)
    # per-ray intervals, including closed-interval edges
    tmin = np.full(len(o), 0.5)
    tmax = g.uniform(1.0, 8.0, len(o))
    compare(rt.SceneDesc.from_world(rt.BvhNode(lst)), o, d, time=tm, tmin=tmin, tmax=tmax, min_hits=500)


def test_book1_scene_rays():
    world, lights, cam = scenes.book1_final(seed=0, moving=True)
    desc = rt.SceneDesc.from_world(world, lights)
    o, d, tm = rays_for([-12, 12, 0, 2.5, -12, 12], 60000, seed=8, spread=0.1)
    compare(desc, o, d, time=tm, min_hits=20000)


def test_cornell_box_instances_and_ties():
    world, lights, cam = scenes.cornell_box()
    desc = rt.SceneDesc.from_world(world, lights)
    o, d, tm = scenes.incoherent_rays(60000, [1, 554, 1, 554, 1, 554], seed=9)
    compare(desc, o, d, time=tm, min_hits=50000)
    # adversarial: axis-parallel rays, origins on walls, rays along box faces and through box corners
    g = np.random.default_rng(10)
    axes = np.eye(3)[g.integers(0, 3, 6000)] * g.choice([-1.0, 1.0], (6000, 1))
    oo = g.uniform(0, 555, (6000, 3))
    oo[:2000, 0] = 0.0
    oo[2000:3000, 1] = 555.0
    compare(desc, oo, axes, min_hits=3000)
    corners = np.array([[265.0, 0, 295], [265.0, 330, 295], [343.0, 554, 332], [213.0, 554, 227], [555.0, 555, 555], [0.0, 0, 0]])
    tgt = corners[g.integers(0, len(corners), 6000)]
    src = g.uniform(50, 500, (6000, 3))
    compare(desc, src, tgt - src, min_hits=3000)


def test_bvh_of_boxes_quad_runs():
    """A BVH whose leaves are boxes (lists of six consecutive quads): the ordered traversal enters them without
    reading the list, per lane for incoherent rays and through the scalar path when a whole wave stands at the
    same box (identical rays)."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    boxes = rt.HittableList()
    g = np.random.default_rng(21)
    for ix in range(8):
        for iz in range(8):
            h = float(g.uniform(0.2, 2.0))
            boxes.add(rt.box_new(rt.Point3(ix, 0.0, iz), rt.Point3(ix + 0.9, h, iz + 0.9), m))
    world = rt.HittableList.with_object(rt.BvhNode(boxes))
    world.add(rt.Sphere(rt.Point3(4.0, 4.0, 4.0), 1.0, m))
    desc = rt.SceneDesc.from_world(world)
    o, d, tm = scenes.incoherent_rays(40000, [0, 8, 0, 5, 0, 8], seed=22)
    for flags in (0, F.RT_HIT_ORDERED):
        compare(desc, o, d, time=tm, min_hits=15000, flags=flags)
        oo = np.tile([[4.3, 6.0, 3.7]], (4096, 1))          # every lane of every wave traces the same ray
        dd = np.tile([[0.1, -1.0, 0.05]], (4096, 1))
        compare(desc, oo, dd, min_hits=4096, flags=flags)


def test_book2_scene_with_media():
    world, lights, cam = scenes.book2_final(seed=0)
    desc = rt.SceneDesc.from_world(world, lights)
    o, d, tm = scenes.incoherent_rays(60000, [-200, 600, 0, 560, -200, 600], seed=11)
    a, b = compare(desc, o, d, time=tm, seed=77, min_hits=30000)
    kinds = set((b["prim"][b["prim"] != F.RT_REF_NONE] >> 28).tolist())
    assert {F.RT_REF_SPHERE, F.RT_REF_QUAD, F.RT_REF_MEDIUM} <= kinds
    # camera-like rays toward the fog-free parts
    d2 = np.array([278.0, 278.0, 0.0]) - np.array([478.0, 278.0, -600.0]) + np.random.default_rng(2).normal(0, 120, (30000, 3))
    o2 = np.tile([478.0, 278.0, -600.0], (30000, 1))
    compare(desc, o2, d2, time=tm[:30000], seed=78, min_hits=15000)


def test_medium_inside_the_boundary_of_a_medium():
    """constant_medium.rs:48-104 takes any Hittable as the boundary, another medium or a list holding one included:
    the outer medium's two boundary walks then each draw for the inner one, from the same stream, in the order the
    recursion does."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    inner = rt.ConstantMedium.new_with_color(rt.Sphere(rt.Point3(0, 0, 0), 1.0, m), 1.5, rt.Color(1, 1, 1))
    shell = rt.HittableList()
    shell.add(inner)
    shell.add(rt.Sphere(rt.Point3(0.3, 0, 0), 1.6, rt.Dielectric(1.5)))
    outer = rt.ConstantMedium.new_with_color(shell, 0.8, rt.Color(.2, .4, .9))
    world = rt.HittableList()
    world.add(outer)
    world.add(rt.Sphere(rt.Point3(0, -101.5, 0), 100.0, m))
    world.add(rt.ConstantMedium.new_with_color(rt.Translate(rt.RotateY(inner, 20.0), rt.Vec3(3.0, 0, 0)), 2.0, rt.Color(1, 1, 1)))
    desc = rt.SceneDesc.from_world(world)
    o, d, tm = scenes.incoherent_rays(30000, [-3, 5, -2, 2, -3, 3], seed=51)
    for flags in (0, F.RT_HIT_ORDERED):
        res = compare(desc, o, d, time=tm, seed=9, min_hits=5000, flags=flags)
    lst = rt.HittableList()
    for k in range(40):
        lst.add(rt.Sphere(rt.Point3(0.4 * k - 8.0, 0.0, 0.0), 0.15, m))
    lst.add(outer)
    compare(rt.SceneDesc.from_world(rt.BvhNode(lst)), o, d, time=tm, seed=10, min_hits=3000, flags=F.RT_HIT_ORDERED)


def test_medium_hits_report_the_u_v_their_record_held():
    """A phase function over an image texture reads u,v that ConstantMedium::hit never writes: they are those of the
    primitive that last wrote the same record -- the list's own temp_rec (hittable_list.rs:46-57), the caller's record
    under BvhNode, Translate and RotateY -- or 0, 0.  Checked here on the closest-hit records themselves."""
    img = rt.ImageTexture(scenes.procedural_image())
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    fog = lambda b, dens: rt.ConstantMedium.new_with_texture(b, dens, img)
    o, d, tm = scenes.incoherent_rays(20000, [-4, 4, -2, 2, -4, 4], seed=61)

    # (1) a list: spheres hit before the medium in list order leave their u,v in temp_rec; spheres after it do not
    lst = rt.HittableList()
    lst.add(rt.Sphere(rt.Point3(0, 0, 0), 0.5, m))
    lst.add(rt.Sphere(rt.Point3(2.5, 0, 0), 0.7, m))
    lst.add(fog(rt.Sphere(rt.Point3(0.5, 0, 0), 2.0, m), 0.6))
    lst.add(rt.Sphere(rt.Point3(-2.5, 0, 0), 0.7, m))
    lst.add(rt.Quad(rt.Point3(-4, -1.9, -4), rt.Vec3(8, 0, 0), rt.Vec3(0, 0, 8), m))
    a, b = compare(rt.SceneDesc.from_world(lst), o, d, time=tm, seed=3, min_hits=5000, medium_uv=True)
    med = (b["prim"] >> 28) == F.RT_REF_MEDIUM
    assert med.sum() > 1000 and (b["u"][med] != 0).sum() > 100 and (b["u"][med] == 0).sum() > 100  # both kinds occur

    # (2) the same members in a BVH (one record handed down, reference order because of the medium), nested in a list,
    #     next to an instanced copy and a list holding a list
    inner = rt.HittableList()
    inner.add(rt.Sphere(rt.Point3(0, 1.2, 0), 0.4, m))
    inner.add(fog(rt.Sphere(rt.Point3(0, 1.2, 0), 1.0, m), 1.5))
    world = rt.HittableList()
    world.add(rt.BvhNode(lst))
    world.add(rt.Translate(rt.RotateY(inner, 30.0), rt.Vec3(0.0, 0.0, 2.5)))
    outer = rt.HittableList()
    outer.add(inner)
    outer.add(rt.Sphere(rt.Point3(0, -1.2, -2.5), 0.5, m))
    world.add(rt.Translate(outer, rt.Vec3(0.0, 0.0, -2.5)))
    for flags in (0, F.RT_HIT_ORDERED):
        a, b = compare(rt.SceneDesc.from_world(world), o, d, time=tm, seed=4, min_hits=5000, flags=flags, medium_uv=True)
    med = (b["prim"] >> 28) == F.RT_REF_MEDIUM
    assert med.sum() > 1000 and (b["u"][med] != 0).sum() > 50

    # (3) a medium whose boundary holds such a medium: the boundary walks use records of their own
    shell = rt.HittableList()
    shell.add(inner)
    shell.add(rt.Sphere(rt.Point3(0, 1.2, 0), 1.6, rt.Dielectric(1.5)))
    w3 = rt.HittableList()
    w3.add(rt.Sphere(rt.Point3(0, -1.0, 0), 0.8, m))
    w3.add(fog(shell, 0.7))
    compare(rt.SceneDesc.from_world(w3), o, d, time=tm, seed=5, min_hits=1000, flags=F.RT_HIT_ORDERED, medium_uv=True)

    # (4) a one-object span stores its object as both children (bvh.rs:41-44), so the list below is walked twice, each
    #     time with a fresh temp_rec: the second time the quad is accepted again at the t the record already holds and
    #     the medium draws again -- a medium hit of the second walk reports the quad's u,v
    dup = rt.HittableList()
    dup.add(rt.Quad(rt.Point3(-3, -3, 1.5), rt.Vec3(6, 0, 0), rt.Vec3(0, 6, 0), m))
    dup.add(fog(rt.Sphere(rt.Point3(0, 0, 0), 1.4, m), 0.5))
    one = rt.HittableList()
    one.add(dup)
    w4 = rt.HittableList()
    w4.add(rt.BvhNode(one))
    for flags in (0, F.RT_HIT_ORDERED):
        a, b = compare(rt.SceneDesc.from_world(w4), o, d, time=tm, seed=6, min_hits=3000, flags=flags, medium_uv=True)
    med = (b["prim"] >> 28) == F.RT_REF_MEDIUM
    assert med.sum() > 500 and (b["u"][med] != 0).sum() > 100


def test_cornell_smoke_media_inside_bvh():
    world, lights, cam = scenes.cornell_smoke()
    desc = rt.SceneDesc.from_world(world, lights)
    o, d, tm = scenes.incoherent_rays(60000, [1, 554, 1, 554, 1, 554], seed=12)
    a, b = compare(desc, o, d, time=tm, seed=5, min_hits=40000)
    assert ((b["prim"] >> 28) == F.RT_REF_MEDIUM).sum() > 500


def test_triangle_mesh_and_objmodel():
    v = scenes.triangle_soup(20000, seed=13, edge=0.05)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    o, d, tm = scenes.incoherent_rays(50000, [0, 1, 0, 1, 0, 1], seed=14)
    a, b = compare(rt.SceneDesc.from_world(rt.Mesh(v, m)), o, d, min_hits=20000)
    a2, b2 = compare(rt.SceneDesc.from_world(rt.ObjModel(v[:600], m)), o[:4000], d[:4000], min_hits=100)
    ico = scenes.icosphere_mesh(subdiv=3)
    compare(rt.SceneDesc.from_world(rt.Mesh(ico, m)), o, d, min_hits=10000)
    oo, dd, _ = scenes.primary_rays(128, [0, 1, 0, 1, 0, 1])
    compare(rt.SceneDesc.from_world(rt.Mesh(ico, m)), oo, dd, min_hits=3000)


def test_ordered_traversal_matches_reference_order():
    """RT_HIT_ORDERED (wide f32 nodes, nearer child first) against the oracle's reference-order result."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    v = scenes.triangle_soup(40000, seed=21, edge=0.04)
    o, d, tm = scenes.incoherent_rays(100000, [0, 1, 0, 1, 0, 1], seed=22)
    compare(rt.SceneDesc.from_world(rt.Mesh(v, m)), o, d, min_hits=50000, flags=F.RT_HIT_ORDERED)
    world, lights, cam = scenes.book1_final(seed=0, moving=True)
    o2, d2, tm2 = rays_for([-12, 12, 0, 2.5, -12, 12], 60000, seed=8, spread=0.1)
    compare(rt.SceneDesc.from_world(world, lights), o2, d2, time=tm2, min_hits=20000, flags=F.RT_HIT_ORDERED)
    for build in (lambda: scenes.book2_final(seed=0), scenes.cornell_smoke, scenes.cornell_box, scenes.quads):
        world, lights, cam = build()
        desc = rt.SceneDesc.from_world(world, lights)
        b = rt.Scene(desc, host_only=True).info().world_bounds
        o3, d3, tm3 = rays_for([b[k] for k in range(6)] if np.isfinite(b[0]) and abs(b[0]) < 1e4 else [-200, 600, 0, 560, -200, 600],
                               40000, seed=23, spread=0.05)
        compare(desc, o3, d3, time=tm3, seed=9, min_hits=3000, flags=F.RT_HIT_ORDERED)
    # the exact-box variant of the ordered traversal (children ordered by the split axis), incl. media and instances
    for build in (lambda: scenes.book2_final(seed=0), scenes.cornell_smoke, scenes.cornell_box):
        world, lights, cam = build()
        desc = rt.SceneDesc.from_world(world, lights)
        o3, d3, tm3 = rays_for([-200, 600, 0, 560, -200, 600], 30000, seed=24, spread=0.05)
        compare(desc, o3, d3, time=tm3, seed=9, min_hits=3000, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
    compare(rt.SceneDesc.from_world(rt.Mesh(v, m)), o, d, min_hits=50000, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
    # GPU-side counters: the ordered traversal must visit far fewer boxes than the reference order
    desc = rt.SceneDesc.from_world(rt.Mesh(v, m))
    g = rt.Scene(desc)
    g.closest_hit(o, d, flags=F.RT_HIT_COUNTERS)
    ref = g.counters()
    ctr = {}
    O.OracleScene(desc).closest_hit(o, d, counters=ctr)
    # same boxes tested as the oracle's recursion; the one-object-span duplicate test (bvh.rs:41-44) is skipped
    assert ref.node_visits == ctr["node_visits"] and ref.prim_tests <= ctr["prim_tests"]
    g.reset_counters()
    g.closest_hit(o, d, flags=F.RT_HIT_COUNTERS | F.RT_HIT_ORDERED)
    fast = g.counters()
    assert fast.node_visits < 0.8 * ref.node_visits and fast.prim_tests < 0.5 * ref.prim_tests
    g.reset_counters()
    g.closest_hit(o, d, flags=F.RT_HIT_COUNTERS | F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
    cull = g.counters()
    assert cull.node_visits < 0.7 * ref.node_visits, (cull.node_visits, ref.node_visits)


def test_degenerate_batches():
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    s = rt.Scene(rt.HittableList.with_object(rt.Sphere(rt.Point3(0, 0, -1), 0.5, m)))
    h = s.closest_hit(np.zeros((0, 3)), np.zeros((0, 3)))
    assert len(h["t"]) == 0
    # ragged against the 64-lane chunking and against the per-wave work ranges of grids of 1..8 workgroups
    for n in (1, 63, 64, 65, 257, 1500, 1793, 3000):
        o = np.zeros((n, 3))
        d = np.tile([0, 0, -1.0], (n, 1))
        d[::2] = [0, 0, 1.0]
        h = s.closest_hit(o, d)
        assert np.array_equal(h["prim"] != F.RT_REF_NONE, np.arange(n) % 2 == 1)
        assert np.all(np.isinf(h["t"][::2]))
    # zero direction components / zero direction: NaN slabs must behave as in aabb.rs:82-107
    lst = rt.HittableList()
    for x in range(8):
        lst.add(rt.Sphere(rt.Point3(x, 0, 0), 0.4, m))
    desc = rt.SceneDesc.from_world(rt.BvhNode(lst))
    o = np.array([[-2, 0, 0], [-2, 0, 0], [3, 5, 0], [3, 0.4, 0], [1, 1, 1]], float)
    d = np.array([[1, 0, 0], [0, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, 0]], float)
    compare(desc, o, d, min_hits=2)


def _edge_rays(bounds, n_grid=12):
    """Axis-parallel rays (zero direction components: the slab test sees 0 * inf), rays starting exactly on
    box planes of a regular grid, diagonal rays through corners, and a zero direction."""
    lo = np.array([bounds[0], bounds[2], bounds[4]])
    hi = np.array([bounds[1], bounds[3], bounds[5]])
    g = np.linspace(0.0, 1.0, n_grid)
    o, d = [], []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            for a in g:
                for b in g:
                    p = np.array([a, b, a])
                    p[ax] = -0.25 if sgn > 0 else 1.25
                    p[(ax + 1) % 3] = a
                    p[(ax + 2) % 3] = b
                    o.append(lo + p * (hi - lo))
                    e = np.zeros(3)
                    e[ax] = sgn
                    d.append(e)
    for a in g:  # diagonals and rays in a coordinate plane
        o.append(lo + np.array([a, 0.0, 0.0]) * (hi - lo)); d.append(np.array([0.0, 1.0, 1.0]))
        o.append(lo + np.array([1.0, a, 1.0]) * (hi - lo)); d.append(np.array([-1.0, 0.0, -1.0]))
        o.append(lo - 0.1 * (hi - lo)); d.append((hi - lo) * np.array([1.0, a + 0.01, 1.0]))
    o.append(0.5 * (lo + hi)); d.append(np.zeros(3))
    return np.array(o), np.array(d)


def test_primitive_bvh_kernel_edge_cases():
    """The dedicated kernel for BVH-only worlds, both node formats: ragged batch sizes around its per-wave
    work ranges, axis-parallel and plane-grazing rays, caller-supplied intervals with negative tmin."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    soup = rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(3000, seed=31, edge=0.08), m))  # 8-bit nodes accepted
    lst = rt.HittableList()
    lst.add(rt.Sphere(rt.Point3(0.5, -100.0, 0.5), 100.0, m))  # dwarfs the rest: f32 nodes are kept
    rng = np.random.default_rng(5)
    for c in rng.uniform(0.0, 1.0, (300, 3)):
        lst.add(rt.Sphere(rt.Point3(*c), 0.03, m))
    spheres = rt.SceneDesc.from_world(rt.BvhNode(lst))
    assert rt.Scene(soup, host_only=True).wide_nodes()[0]["has_box8"] == 1
    assert rt.Scene(spheres, host_only=True).wide_nodes()[0]["has_box8"] == 0
    for desc, min_hits in ((soup, 300), (spheres, 300)):
        o, d = _edge_rays([0, 1, 0, 1, 0, 1])
        compare(desc, o, d, min_hits=min_hits, flags=F.RT_HIT_ORDERED)
        oo, dd, tm = scenes.incoherent_rays(5000, [0, 1, 0, 1, 0, 1], seed=32)
        g = np.random.default_rng(6)
        tmin = g.uniform(-0.5, 0.05, len(oo))   # mostly negative: hits behind the origin count
        tmax = g.uniform(0.05, 0.6, len(oo))
        compare(desc, oo, dd, time=tm, tmin=tmin, tmax=tmax, min_hits=500, flags=F.RT_HIT_ORDERED)
        for n in (1, 2, 63, 64, 65, 129, 1000):
            compare(desc, oo[:n], dd[:n], min_hits=0, flags=F.RT_HIT_ORDERED)


def test_packed_and_64_byte_nodes_agree(monkeypatch):
    """A triangle mesh is traced on the 48-byte packed nodes (children addressed from two base indices, triangles in
    leaf order); MI355RT_PACKED=0 keeps the 64-byte nodes.  Same boxes, so the same (t, primitive), also with caller
    intervals that only bound t from above (the packed kernel's sort keys need tmin >= 0)."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    desc = rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(20000, seed=41, edge=0.05), m))
    o, d, tm = scenes.incoherent_rays(40000, [0, 1, 0, 1, 0, 1], seed=42)
    tmax = np.random.default_rng(7).uniform(0.05, 2.0, len(o))
    packed = rt.Scene(desc)
    monkeypatch.setenv("MI355RT_PACKED", "0")
    plain = rt.Scene(desc)
    monkeypatch.delenv("MI355RT_PACKED")
    for kw in ({}, {"tmax": tmax}):
        a = packed.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_COUNTERS, **kw)
        b = plain.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_COUNTERS, **kw)
        assert np.array_equal(a["t"], b["t"]) and (a["prim"] != -1).sum() > 10000
        same = a["prim"] == b["prim"]
        assert same.mean() > 0.9999  # an exact tie between two triangles may go to either
        for k in ("u", "v", "front_face", "mat"):
            assert np.array_equal(a[k][same], b[k][same]), k
    ca, cb = packed.counters(), plain.counters()
    # Not a parity bound (t above is compared exactly): it guards against one of the two layouts losing children, i.e.
    # walking a different tree.  The counts cannot be equal: children with equal entry distances may be visited in
    # another order, and the packed kernel lets a lane walk on past a triangle it has reached until enough lanes have
    # one to test, so it prunes a little later -- 1 to 3 % more node visits, depending on how the waves fill, which the
    # refill and quorum settings change.  A lost or duplicated subtree would move the count by far more.
    assert abs(int(ca.node_visits) - int(cb.node_visits)) <= 0.03 * cb.node_visits and ca.prim_tests > 0
    compare(desc, o[:5000], d[:5000], min_hits=1000, flags=F.RT_HIT_ORDERED)
    packed.close(); plain.close()


def test_prepared_rays_of_the_packed_tree(monkeypatch):
    """Batches of 2^18 rays or more are binned by origin cell and handed to the packed kernel as 64-byte prepared
    records (origin, direction, f32 inverse, index); the kernel derives the offsets of its conservative slab test from
    the inverse.  With the threshold lowered the same path takes small batches: axis-parallel rays and a zero
    direction (the inverse is infinite, 0 * inf in the offsets), origins on box planes, ragged sizes, an upper end
    of the interval read from the caller's array -- all against the oracle."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    desc = rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(6000, seed=51, edge=0.07), m))
    monkeypatch.setenv("MI355RT_REORDER_MIN", "1")
    monkeypatch.setenv("MI355RT_REORDER", "2")  # also where the pass would not pay
    o, d = _edge_rays([0, 1, 0, 1, 0, 1])
    compare(desc, o, d, min_hits=300, flags=F.RT_HIT_ORDERED)
    oo, dd, tm = scenes.incoherent_rays(30000, [-0.2, 1.2, -0.2, 1.2, -0.2, 1.2], seed=52)
    compare(desc, oo, dd, time=tm, min_hits=5000, flags=F.RT_HIT_ORDERED)
    tmax = np.random.default_rng(8).uniform(0.02, 1.0, len(oo))
    compare(desc, oo, dd, tmax=tmax, min_hits=2000, flags=F.RT_HIT_ORDERED)
    for n in (1, 63, 65, 1000):
        compare(desc, oo[:n], dd[:n], min_hits=0, flags=F.RT_HIT_ORDERED)
    scene = rt.Scene(desc)
    scene.closest_hit(oo, dd, flags=F.RT_HIT_ORDERED)
    assert scene.counters().ms_last_reorder > 0.0  # the batch did go through the reorder pass
    scene.close()


def test_deep_stacks_of_instanced_nested_lists():
    """A BVH over Translate(RotateY(nested lists)) leaves: the ordered traversal keeps deferred siblings, restore
    entries and list continuations on the LDS stack at the same time (up to 23 entries with 4-wide nodes here, 27
    with the binary nodes the deeper scene falls back to).  Both visit orders against the oracle."""
    from test_host_flatten import _instanced_leaf
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    for n_leaves, nest in ((256, 2), (256, 8)):
        lst = rt.HittableList()
        for x in range(n_leaves):
            lst.add(_instanced_leaf(m, x, nest))
        desc = rt.SceneDesc.from_world(rt.HittableList.with_object(rt.BvhNode(lst)))
        o, d, tm = scenes.incoherent_rays(30000, [0, 256, -0.5, 1.5, -0.5, 1.0], seed=61)
        d[:, 0] *= 8.0  # along the row of instances: many boxes per ray
        for flags in (0, F.RT_HIT_ORDERED):
            compare(desc, o, d, time=tm, min_hits=3000, flags=flags)


def _tangent_rays():
    """Triangles, quads and spheres on integer coordinates in a BVH, and rays whose entry into a bounding box and
    exit from it coincide exactly (every coordinate and parameter is a small dyadic rational, so the slab
    arithmetic is exact): through box edges, through box corners, starting on box planes."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    lst = rt.HittableList()
    cells = []
    for ix in range(4):
        for iy in range(4):
            for iz in range(2):
                b = np.array([4.0 * ix, 4.0 * iy, 4.0 * iz])
                cells.append(b)
                k = (ix + iy + iz) % 3
                if k == 0:   # triangle with its vertices on the corners of its box
                    lst.add(rt.Triangle(rt.Point3(*b), rt.Point3(*(b + [2, 0, 1])), rt.Point3(*(b + [0, 2, 2])), m))
                elif k == 1:  # tilted quad: corners on box corners
                    lst.add(rt.Quad(rt.Point3(*b), rt.Vec3(2, 0, 1), rt.Vec3(0, 2, 0), m))
                else:
                    lst.add(rt.Sphere(rt.Point3(*(b + 1.0)), 1.0, m))
    o, d = [], []
    for b in cells:
        lo, hi = b, b + np.array([2.0, 2.0, 2.0])
        for corner in ((lo[0], lo[1], lo[2]), (hi[0], lo[1], lo[2] + 1.0), (lo[0], hi[1], hi[2]), (hi[0], hi[1], hi[2])):
            c = np.array(corner)
            for dirv in ((1, -1, 0), (-1, 1, 0), (1, 1, -1), (1, -1, 1), (-1, -1, 1), (1, 0, -1), (0, 1, 1), (2, -1, 0), (1, -2, 1)):
                dv = np.array(dirv, float)
                for back in (1.0, 2.0, 0.0):  # back = 0: the origin sits on the corner / edge itself
                    o.append(c - back * dv); d.append(dv)
        # through the middle of box edges, in the plane perpendicular to the edge
        for e0, e1 in (((lo[0], lo[1]), (1, -1)), ((hi[0], lo[1]), (1, 1)), ((hi[0], hi[1]), (-1, 1)), ((lo[0], hi[1]), (1, 1))):
            for z in (lo[2] + 0.5, lo[2] + 1.0):
                o.append(np.array([e0[0] - 2.0 * e1[0], e0[1] - 2.0 * e1[1], z])); d.append(np.array([e1[0], e1[1], 0.0]))
    return rt.SceneDesc.from_world(rt.BvhNode(lst)), np.array(o), np.array(d)


def test_rays_tangent_to_boxes():
    """aabb.rs:102 culls a box when the clipped interval is empty OR a single point (`ray_t.max <= ray_t.min`): a ray
    that enters a box exactly where it leaves it (edge, corner) never reaches what is inside, even a primitive
    that touches that very edge.  RT_HIT_REFERENCE_ORDER keeps that rule and must equal the oracle ray for ray.
    RT_HIT_ORDERED tests conservative boxes (entry <= exit accepts) and therefore still tests such a primitive:
    it may report a hit, geometrically on the primitive, where the reference reports none or a farther one.
    That is a deviation beyond the tie rule, confined to rays whose box entry and exit coincide exactly; this
    test pins it (which rays differ, and that every differing hit is real)."""
    desc, o, d = _tangent_rays()
    gpu, cpu = rt.Scene(desc), O.OracleScene(desc)
    want = cpu.closest_hit(o, d)
    exact = gpu.closest_hit(o, d, flags=F.RT_HIT_REFERENCE_ORDER)
    assert np.array_equal(exact["prim"], want["prim"]) and np.array_equal(exact["t"], want["t"])
    fast = gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED)
    assert (want["prim"] != F.RT_REF_NONE).sum() > 200
    differ = np.nonzero(fast["t"] != want["t"])[0]
    # never a hit lost or pushed back: the conservative traversal only ever finds more
    assert np.all(fast["t"][differ] < want["t"][differ])
    for i in differ:  # every extra hit lies on its primitive, and the exact-box traversal of the same ray skipped a box there
        p = o[i] + fast["t"][i] * d[i]
        assert np.allclose(p, [fast["px"][i], fast["py"][i], fast["pz"][i]], rtol=0, atol=1e-12)
        kind = int(fast["prim"][i]) >> 28
        assert kind in (F.RT_REF_TRI, F.RT_REF_QUAD, F.RT_REF_SPHERE)
        if kind == F.RT_REF_SPHERE:
            s = desc.arrays["spheres"][int(fast["prim"][i]) & 0x0FFFFFFF]
            assert abs(np.linalg.norm(p - s["center"]) - s["radius"]) < 1e-9
        else:
            assert 0.0 <= fast["u"][i] <= 1.0 and 0.0 <= fast["v"][i] <= 1.0
    # the batch is built to provoke it (about a quarter of these rays); it vanishes once the rays are nudged off the edges
    assert 0 < len(differ) <= 0.4 * len(o), len(differ)
    # RT_HIT_EXACT_CULL: near-side-first order on the exact boxes, the reference's cull rule intact: same t as the oracle on
    # every ray, the primitive differing only at exactly equal t
    cull = gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
    assert np.array_equal(cull["t"], want["t"])
    assert np.mean(cull["prim"] != want["prim"]) < 0.05
    # ... and on rays a hair off the edges, on either side of what f32 slabs can tell apart: the exact-cull traversal takes a
    # child on the strength of the f32 test alone only when that test passes with room to spare
    for eps in (1e-13, 1e-10, 1e-8, 1e-7, 1e-6, 1e-5):
        for sign in (1.0, -1.0):
            oe = o + sign * eps * np.array([0.37, -0.21, 0.13])
            ce, we = gpu.closest_hit(oe, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL), cpu.closest_hit(oe, d)
            assert np.array_equal(ce["t"], we["t"]), (eps, sign)
    o2 = o + 1e-3 * np.array([0.37, -0.21, 0.13])
    a, b = gpu.closest_hit(o2, d, flags=F.RT_HIT_ORDERED), cpu.closest_hit(o2, d)
    assert np.array_equal(a["t"], b["t"])
    gpu.close()


def test_obj_file_to_gpu(tmp_path):
    """File -> rt_obj_load -> scene description -> HIP kernels, against the oracle on the same description:
    Mesh::from_obj (BVH over the file's triangles, mesh.rs:101-167) in both visit orders, and ObjModel::load with
    scale and offset (linear scan, obj_loader.rs:21-101)."""
    from test_host_flatten import _write_obj
    p = str(tmp_path / "lumpy.obj")
    _write_obj(p, np.random.default_rng(3), n_u=40, n_v=24)
    m = rt.Metal(rt.Color(.8, .8, .9), 0.0)
    mesh = rt.Mesh.from_obj(p, m)
    assert len(mesh.vertices) > 2000
    o, d, tm = scenes.incoherent_rays(40000, [-1.3, 1.3, -1.3, 1.3, -1.3, 1.3], seed=71)
    for flags in (0, F.RT_HIT_ORDERED, F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL):
        compare(rt.SceneDesc.from_world(mesh), o, d, min_hits=15000, flags=flags)
    model = rt.ObjModel.load(p, m, 2.5, rt.Point3(10.0, -3.0, 0.125))
    o2 = o[:6000] * 2.5 + np.array([10.0, -3.0, 0.125])
    compare(rt.SceneDesc.from_world(model), o2, d[:6000], min_hits=2000)
    # the model inside a scene graph: instanced, next to other members
    world = rt.HittableList.with_object(rt.Translate(rt.RotateY(mesh, 30.0), rt.Vec3(0.5, 0.0, 0.0)))
    world.add(rt.Sphere(rt.Point3(0.0, -101.5, 0.0), 100.0, rt.Lambertian(rt.Color(.5, .5, .5))))
    compare(rt.SceneDesc.from_world(world), o, d, min_hits=15000, flags=F.RT_HIT_ORDERED)


def test_rays_at_shared_edges_and_vertices_from_far_away(monkeypatch):
    """Rays aimed exactly at the vertices and edge midpoints of a closed mesh, from origins up to 10^5 scene sizes away:
    the triangles around a vertex are hit within a few f32 units in the last place of each other (or at the same t), and
    their boxes are entered at nearly the same distance.  The entry distance a deferred subtree is stored with must
    stay a lower bound, or the pop-time cull drops the subtree that holds the closer triangle.
    The yardstick is the closest hit over ALL triangles, i.e. the oracle's linear scan of the same triangles (ObjModel,
    obj_loader.rs:86-101): such rays touch triangle boxes at single points, which Aabb::hit culls (aabb.rs:102) and the
    conservative boxes of the ordered traversal do not, so the reference's BVH itself misses some of these hits (the
    documented deviation of RT_HIT_ORDERED); the ordered traversal must not miss any, with batches traced as they come
    and through the binning pass."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    tris = scenes.icosphere_mesh(subdiv=4)
    desc = rt.SceneDesc.from_world(rt.Mesh(tris, m))
    scan = O.OracleScene(rt.SceneDesc.from_world(rt.ObjModel(tris, m)))
    bvh = O.OracleScene(desc)
    g = np.random.default_rng(77)
    verts = tris.reshape(-1, 3)
    mids = 0.5 * (tris[:, 0] + tris[:, 1])
    targets = np.concatenate([verts[g.integers(0, len(verts), 3000)], mids[g.integers(0, len(mids), 3000)]])
    dirs = g.normal(size=targets.shape)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    for pass_on in (False, True):
        if pass_on:
            monkeypatch.setenv("MI355RT_REORDER_MIN", "1")
            monkeypatch.setenv("MI355RT_REORDER", "2")
        gpu = rt.Scene(desc)
        for dist, unit in ((3.0, False), (1e3, True), (1e5, True), (1e5, False)):
            o = targets - dirs * dist
            d = dirs if unit else (targets - o)  # unit directions: t ~ dist; unnormalised: t ~ 1
            a = gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED)
            b = scan.closest_hit(o, d, nthreads=8)
            assert (b["prim"] != F.RT_REF_NONE).sum() >= 5500
            assert np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64)), (dist, unit, pass_on)
            same = a["prim"] == b["prim"]  # several triangles meet at every target: any of those tied at the closest t may be reported
            for f in ("u", "v", "front_face", "mat", "nx", "ny", "nz"):
                assert np.array_equal(a[f][same], b[f][same]), f
            # RT_HIT_EXACT_CULL on the same rays: the reference's own answer (its BVH culls boxes touched at a point,
            # so some of the hits above are missed there too), also where f32 slabs cannot tell touching from crossing
            c = gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL)
            w = bvh.closest_hit(o, d, nthreads=8)
            # every ray here goes through a point where five or six triangles meet, so their hits lie within rounding of each
            # other and a box can begin, in floating point, just behind a neighbour's hit: which of them is kept then depends
            # on the visit order.  No hit appears or disappears, and t agrees to rounding; on nearly all rays exactly.
            assert np.array_equal(c["prim"] == F.RT_REF_NONE, w["prim"] == F.RT_REF_NONE)
            hit = w["prim"] != F.RT_REF_NONE
            assert np.all(np.abs(c["t"][hit] - w["t"][hit]) <= 1e-12 * np.abs(w["t"][hit]))
            assert np.mean(c["t"][hit] != w["t"][hit]) < 0.02
        gpu.close()
    # the f32 short cut of the exact-cull traversal decides nothing differently: every output is the same with it off
    o, d = targets - dirs * 1e3, dirs
    res = []
    for sure in ("1", "0"):
        monkeypatch.setenv("MI355RT_EXACT_SURE", sure)
        gpu = rt.Scene(desc)
        res.append(gpu.closest_hit(o, d, flags=F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL))
        gpu.close()
    for f in res[0]:
        assert np.array_equal(res[0][f].view(np.uint8), res[1][f].view(np.uint8)), f


def test_exact_cull_on_f32_nodes_with_and_without_the_short_cut(monkeypatch):
    """RT_HIT_ORDERED | RT_HIT_EXACT_CULL on a mesh walks the collapse of the reference's tree on f32 nodes and asks
    Aabb::hit's f64 arithmetic (aabb.rs:82-107) only where the f32 slabs leave room for doubt.  On a soup of 2^17
    triangles: equal to the oracle on a sample, and byte for byte the same outputs over 2^21 rays whether the short cut
    is taken or every accepted child is confirmed in f64 (MI355RT_EXACT_SURE=0), traced as they come and through the
    binning pass, and on axis-parallel rays, whose infinite 1/d the f32 form cannot bound."""
    desc = rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 17, seed=2))
    o, d, _ = scenes.incoherent_rays(1 << 21, [0, 1, 0, 1, 0, 1], seed=5)
    d[::7, 0] = 0.0          # axis-parallel in x
    d[::11, 1:] = 0.0        # along the x axis
    flags = F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL
    res = {}
    for sure in ("1", "0"):
        monkeypatch.setenv("MI355RT_EXACT_SURE", sure)
        gpu = rt.Scene(desc)
        res[sure, "pass"] = gpu.closest_hit(o, d, flags=flags)             # 2^21 >= the binning pass's threshold
        res[sure, "plain"] = gpu.closest_hit(o[:100000], d[:100000], flags=flags)
        gpu.close()
    for leg in ("pass", "plain"):
        for f in res["1", leg]:
            assert np.array_equal(res["1", leg][f].view(np.uint8), res["0", leg][f].view(np.uint8)), (leg, f)
    want = O.OracleScene(desc).closest_hit(o[:30000], d[:30000], nthreads=8)
    assert (want["prim"] != F.RT_REF_NONE).sum() > 5000
    for leg, n in (("pass", 30000), ("plain", 30000)):
        got = res["1", leg]
        assert np.array_equal(got["t"][:n].view(np.uint64), want["t"].view(np.uint64)), leg
        assert np.mean(got["prim"][:n] != want["prim"]) < 0.01


def test_traversal_only_tree_on_awkward_meshes():
    """Meshes the surface-area heuristic and the spatial splits have to cope with, each large enough (>= 1024
    triangles) to get the traversal-only tree, each against the oracle in the nearest-first mode and in the reference's
    order: every triangle the same one (all centres and boxes equal: no split separates anything), a vertex at 1e30 and
    one at 1e308 (beyond f32: that mesh keeps the collapse of the reference's tree), a few triangles as
    large as the scene among thousands of small ones (they straddle every plane), coplanar axis-aligned triangles (boxes
    of zero thickness, padded as aabb.rs:125-136 pads them), needles a thousand times longer than wide, a mesh of
    exactly 1024 triangles and one of 1023 (which keeps the collapse of the reference's tree)."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    g = np.random.default_rng(101)
    soup = scenes.triangle_soup(4000, seed=61, edge=0.05)
    cases = {}
    cases["identical"] = (np.repeat(soup[:1], 2048, axis=0), 0.0)
    mixed = soup.copy()
    mixed[:6] = g.uniform(-0.5, 1.5, (6, 3, 3))
    cases["huge among small"] = (mixed, 0.02)
    flat = soup[:3000].copy()
    flat[:, :, 1] = np.round(flat[:, :1, 1] * 8) / 8          # eight horizontal planes
    cases["coplanar"] = (flat, 0.05)
    a = g.uniform(0, 1, (2500, 3))
    dirs = g.normal(size=(2500, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    side = np.cross(dirs, g.normal(size=(2500, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
    cases["needles"] = (np.stack([a, a + 0.4 * dirs, a + 0.4 * dirs + 4e-4 * side], axis=1), 0.02)
    far = soup.copy()
    far[7, 1, 2] = 1e30                      # a vertex far outside everything: finite in f32, the scene's boxes become useless for 8-bit grids
    cases["one vertex at 1e30"] = (far, 0.02)
    farther = soup.copy()
    farther[7, 1, 2] = 1e308                 # not finite in f32: no traversal-only tree, the scene must still be traced
    cases["one vertex at 1e308"] = (farther, 0.02)
    cases["1024"] = (soup[:1024], 0.02)
    cases["1023"] = (soup[:1023], 0.02)
    o, d, _ = scenes.incoherent_rays(30000, [-0.1, 1.1, -0.1, 1.1, -0.1, 1.1], seed=62)
    aim = soup[g.integers(0, len(soup), 3000)].mean(axis=1)   # some rays aimed at triangle centres
    o[:3000] = aim - d[:3000] * 0.7
    for name, (tri, max_ties) in cases.items():
        desc = rt.SceneDesc.from_world(rt.Mesh(np.ascontiguousarray(tri), m))
        s = rt.Scene(desc, host_only=True)
        assert s.fast_tree_info()[2] == (len(tri) >= 1024 and name != "one vertex at 1e308"), name
        s.close()
        if name == "identical":  # every hit is a 2048-fold tie: only t can be compared, in either order
            gpu, cpu = rt.Scene(desc), O.OracleScene(desc)
            w = g.dirichlet([1.0, 1.0, 1.0], size=5000)
            oi = w @ tri[0] - d[:5000] * 0.7   # towards points of the one triangle
            oi[::10] = o[:5000:10]             # and some that miss it
            want = cpu.closest_hit(oi, d[:5000], nthreads=8)
            assert (want["prim"] != F.RT_REF_NONE).sum() > 4000
            for flags in (0, F.RT_HIT_ORDERED, F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL):
                got = gpu.closest_hit(oi, d[:5000], flags=flags)
                assert np.array_equal(got["t"].view(np.uint64), want["t"].view(np.uint64)), flags
            gpu.close()
            continue
        for flags in (0, F.RT_HIT_ORDERED, F.RT_HIT_ORDERED | F.RT_HIT_EXACT_CULL):
            compare(desc, o, d, min_hits=200, flags=flags, max_ties=max(max_ties, 0.02))


def test_one_scene_from_several_threads():
    """The reference calls world.hit on one world from every rayon worker (camera.rs:99).  Several host threads call
    rt_closest_hit on ONE scene at once (ctypes releases the interpreter lock during the call): the library puts the
    calls in a row, and every thread gets the results of a call made alone.  Device-pointer calls on different streams
    of one scene are ordered on the device the same way."""
    import threading
    world, lights, cam = scenes.book2_final(seed=0)
    desc = rt.SceneDesc.from_world(world, lights)
    gpu = rt.Scene(desc)
    batches = [scenes.incoherent_rays(30000, [-200, 600, 0, 560, -200, 600], seed=90 + k) for k in range(4)]
    alone = [gpu.closest_hit(o, d, time=tm, seed=k, flags=F.RT_HIT_ORDERED) for k, (o, d, tm) in enumerate(batches)]
    got = [None] * len(batches)
    errors = []

    def worker(k):
        try:
            for _ in range(3):
                o, d, tm = batches[k]
                got[k] = gpu.closest_hit(o, d, time=tm, seed=k, flags=F.RT_HIT_ORDERED)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(batches))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b in zip(alone, got):
        for f in a:
            assert np.array_equal(a[f].view(np.uint8), b[f].view(np.uint8)), f
    # device pointers, one stream per batch, no host synchronisation between the calls
    import torch
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in batches]
    outs = []
    for k, (o, d, tm) in enumerate(batches):
        rays = {nm: torch.from_numpy(np.ascontiguousarray(o[:, j])).to(dev) for j, nm in enumerate(("ox", "oy", "oz"))}
        rays.update({nm: torch.from_numpy(np.ascontiguousarray(d[:, j])).to(dev) for j, nm in enumerate(("dx", "dy", "dz"))})
        rays["time"] = torch.from_numpy(np.ascontiguousarray(tm)).to(dev)
        hits = {"t": torch.empty(len(o), dtype=torch.float64, device=dev), "prim": torch.empty(len(o), dtype=torch.int32, device=dev)}
        outs.append((rays, hits))
    torch.cuda.synchronize()
    for k, (rays, hits) in enumerate(outs):
        gpu.closest_hit_device(rays, hits, len(batches[k][0]), seed=k, flags=F.RT_HIT_ORDERED, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k, (rays, hits) in enumerate(outs):
        assert np.array_equal(hits["t"].cpu().numpy().view(np.uint64), alone[k]["t"].view(np.uint64))
        assert np.array_equal(hits["prim"].cpu().numpy().astype(np.uint32), alone[k]["prim"])
    gpu.close()


def test_reserved_scene_allocates_nothing_in_device_calls():
    """rt_scene_reserve sizes the scratch of the *_device entry points ahead of time: calls within the reserved sizes
    leave the free device memory as it was (general traversal: 4 bytes per ray; a triangle mesh whose batches are
    binned: the prepared records)."""
    import torch
    dev = torch.device("cuda", 0)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    world, lights, _ = scenes.cornell_box()
    cases = [(rt.SceneDesc.from_world(world, lights), [1, 554, 1, 554, 1, 554]),
             (rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(20000, seed=13, edge=0.05), m)), [0, 1, 0, 1, 0, 1])]
    n = 300000
    for desc, box in cases:
        gpu = rt.Scene(desc)
        o, d, tm = scenes.incoherent_rays(n, box, seed=43)
        rays = {nm: torch.from_numpy(np.ascontiguousarray(o[:, j])).to(dev) for j, nm in enumerate(("ox", "oy", "oz"))}
        rays.update({nm: torch.from_numpy(np.ascontiguousarray(d[:, j])).to(dev) for j, nm in enumerate(("dx", "dy", "dz"))})
        hits = {"t": torch.empty(n, dtype=torch.float64, device=dev), "prim": torch.empty(n, dtype=torch.int32, device=dev),
                "u": torch.empty(n, dtype=torch.float64, device=dev)}
        gpu.reserve(max_rays=n)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        for flags in (F.RT_HIT_REFERENCE_ORDER, F.RT_HIT_ORDERED):
            gpu.closest_hit_device(rays, hits, n, flags=flags, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before
        assert int((hits["prim"] != -1).sum().item()) > n // 10
        gpu.close()
