"""Random scene graphs against the oracle on the GPU (not collected by pytest: run by name).
   python tests/fuzz_scenes.py [n_scenes] [seed]
Nested lists, BVHs over mixed children (primitives, boxes, instances, media), transform chains; both
traversal orders; coherent (whole waves on the same ray) and incoherent batches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F, scenes
from test_gpu_closest_hit import compare

def rand_prim(g, m):
    k = g.integers(0, 7)
    c = g.uniform(-4, 4, 3)
    if k == 4:
        return rt.Sphere.new_moving(rt.Point3(*c), rt.Point3(*(c + g.uniform(-.7, .7, 3))), float(g.uniform(0.2, 0.8)), m)
    if k == 5:  # a small mesh: a BVH over triangles of its own (mesh.rs:163)
        return rt.Mesh(scenes.triangle_soup(int(g.integers(1, 40)), seed=int(g.integers(1 << 20)), edge=0.8) * 2.0 + c, m)
    if k == 6:  # the same triangles as a flat list (obj_loader.rs:46-75)
        return rt.ObjModel(scenes.triangle_soup(int(g.integers(1, 12)), seed=int(g.integers(1 << 20)), edge=0.8) * 2.0 + c, m)
    if k == 0:
        return rt.Sphere(rt.Point3(*c), float(g.uniform(0.2, 0.8)), m)
    if k == 1:
        return rt.Quad(rt.Point3(*c), rt.Vec3(*g.uniform(-1.5, 1.5, 3)), rt.Vec3(*g.uniform(-1.5, 1.5, 3)), m)
    if k == 2:
        return rt.Triangle(rt.Point3(*c), rt.Point3(*(c + g.uniform(-1.5, 1.5, 3))), rt.Point3(*(c + g.uniform(-1.5, 1.5, 3))), m)
    return rt.box_new(rt.Point3(*c), rt.Point3(*(c + g.uniform(0.2, 1.5, 3))), m)

RICH = {"on": False, "phase": None}  # rich media: any node as a boundary (a medium inside it included), phase functions over an image

def rand_node(g, m, depth, allow_medium, media_level=0):
    if RICH["on"]:
        return rand_node_rich(g, m, depth, allow_medium, media_level)
    r = g.random()
    if depth >= 3 or r < 0.35:
        return rand_prim(g, m)
    if r < 0.55:
        lst = rt.HittableList()
        for _ in range(int(g.integers(1, 7))):
            lst.add(rand_node(g, m, depth + 1, allow_medium))
        return lst
    if r < 0.75:
        lst = rt.HittableList()
        for _ in range(int(g.integers(1, 25))):
            lst.add(rand_node(g, m, depth + 1, False))  # media inside a BVH are covered by the smoke scene; keep trees medium-free here too sometimes
        return rt.BvhNode(lst)
    if r < 0.9:
        child = rand_node(g, m, depth + 1, allow_medium)
        if g.random() < 0.5:
            return rt.Translate(child, rt.Vec3(*g.uniform(-2, 2, 3)))
        return rt.RotateY(child, float(g.uniform(-90, 90)))
    if allow_medium:
        b = rt.Sphere(rt.Point3(*g.uniform(-3, 3, 3)), float(g.uniform(0.8, 2.5)), m) if g.random() < 0.5 else rt.box_new(rt.Point3(-1, -1, -1), rt.Point3(*g.uniform(0, 2, 3)), m)
        return rt.ConstantMedium.new_with_color(b, float(g.uniform(0.05, 0.8)), rt.Color(1, 1, 1))
    return rand_prim(g, m)

def rand_node_rich(g, m, depth, allow_medium, media_level):
    r = g.random()
    if depth >= 3 or r < 0.3:
        return rand_prim(g, m)
    if r < 0.55:
        lst = rt.HittableList()
        for _ in range(int(g.integers(1, 6))):
            lst.add(rand_node_rich(g, m, depth + 1, allow_medium, media_level))
        return lst
    if r < 0.7:
        lst = rt.HittableList()
        for _ in range(int(g.integers(1, 12))):
            lst.add(rand_node_rich(g, m, depth + 1, allow_medium and g.random() < 0.5, media_level))
        return rt.BvhNode(lst)
    if r < 0.8:
        child = rand_node_rich(g, m, depth + 1, allow_medium, media_level)
        if g.random() < 0.5:
            return rt.Translate(child, rt.Vec3(*g.uniform(-2, 2, 3)))
        return rt.RotateY(child, float(g.uniform(-90, 90)))
    if allow_medium and media_level < 2:
        if g.random() < 0.5:
            b = rt.Sphere(rt.Point3(*g.uniform(-3, 3, 3)), float(g.uniform(0.8, 2.5)), m)
        else:  # anything, a medium inside it included (constant_medium.rs:48-104 takes any Hittable)
            b = rand_node_rich(g, m, depth + 1, True, media_level + 1)
        dens = float(g.uniform(0.05, 0.8))
        if RICH["phase"] is not None:
            return rt.ConstantMedium.new_with_texture(b, dens, RICH["phase"])
        return rt.ConstantMedium.new_with_color(b, dens, rt.Color(1, 1, 1))
    return rand_prim(g, m)

failures = []

def describe(node, depth):
    name = type(node).__name__
    extra = ""
    if name == "RotateY": extra = f" {node.angle:.3f}"
    if name == "Translate": extra = f" {node.offset}"
    print("  " * depth + name + extra)
    for attr in ("objects",):
        for c in getattr(node, attr, []) or []:
            describe(c, depth + 1)
    for attr in ("object", "boundary"):
        c = getattr(node, attr, None)
        if c is not None: describe(c, depth + 1)

def run(n_scenes, seed, rich_media=False):
    del failures[:]
    g = np.random.default_rng(seed)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    done = 0
    RICH["on"] = rich_media
    image = rt.ImageTexture(scenes.procedural_image(32, 16))
    for k in range(n_scenes):
        # every other rich scene scatters with an image texture: its medium hits then report the u,v their record held
        RICH["phase"] = image if (rich_media and k % 2 == 0) else None
        uv = RICH["phase"] is not None
        world = rt.HittableList()
        for _ in range(int(g.integers(1, 8))):
            world.add(rand_node(g, m, 0, True))
        try:
            desc = rt.SceneDesc.from_world(world)
            rt.Scene(desc, host_only=True)
        except Exception as e:  # e.g. nesting the library rejects: not what is being tested
            print("scene", k, "rejected:", str(e)[:80]); continue
        o, d, tm = scenes.incoherent_rays(6000, [-5, 5, -5, 5, -5, 5], seed=int(g.integers(1 << 30)))
        oo = np.repeat(o[:40], 128, axis=0); dd = np.repeat(d[:40], 128, axis=0)  # two waves per distinct ray
        for flags in (0, F.RT_HIT_ORDERED):
            for name, (ro, rd, rt_) in (("incoherent", (o, d, tm)), ("coherent", (oo, dd, None))):
                try:
                    compare(desc, ro, rd, time=rt_, min_hits=0, flags=flags, medium_uv=uv)
                except AssertionError as e:
                    print("MISMATCH scene", k, "flags", flags, name, str(e)[:400])
                    describe(world, 0)
                    failures.append(k)
        done += 1
        if done % 100 == 0: print("...", done, "scenes compared,", len(failures), "mismatches", flush=True)
    RICH["on"] = False
    print("fuzz ok:" if not failures else "fuzz FAILED:", done, "scenes", "failures", failures)
    return done, list(failures)


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    done, failed = run(n_scenes, seed, rich_media=len(sys.argv) > 3 and sys.argv[3] == "rich")
    sys.exit(1 if failed else 0)

if __name__ == "__main__":
    main()
