"""Host logic of the product on the CPU: the flattener of libmi355rt.so (RT_SCENE_HOST_ONLY,
no device touched) against the oracle's pointer tree -- bounding boxes and BVH split order
one-to-one -- plus the C ABI surface and its error behaviour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_ffi as O
import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F
from ray_tracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    declared = set(re.findall(r"^(?:int|void|double|const char\*)\s+(rt_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared, "no declarations parsed"
    assert declared == set(F.PROTOTYPES), declared ^ set(F.PROTOTYPES)
    raw = C.CDLL(F.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_struct_sizes_match_header_layout():
    assert C.sizeof(F.rt_bvh_node) == 64
    assert C.sizeof(F.rt_sphere) == 64 and C.sizeof(F.rt_quad) == 80 and C.sizeof(F.rt_triangle) == 80
    assert C.sizeof(F.rt_object) == 48 and C.sizeof(F.rt_camera) == 144


def test_rng_matches_oracle_stream():
    for idx in range(9):
        assert F.lib.rt_rng_uniform(0x1234567890ABCDEF, 1, 7, 9, idx) == O.olib.oracle_rng_uniform(0x1234567890ABCDEF, 1, 7, 9, idx)


def _compare_bvh(desc, obj_id, scene=None):
    """Product pre-order node array vs oracle pre-order dump for the BVH object `obj_id` (`scene`: a scene made from
    `desc` some other way than by the host recursion)."""
    host = scene if scene is not None else rt.Scene(desc, host_only=True)
    orc = O.OracleScene(desc)
    want = orc.bvh_dump(obj_id)
    nodes = host.nodes()
    # locate this BVH's root in the flat array: bounds equal to the object's bounding box and subtree size match
    b = host.object_bounds(obj_id)
    roots = [i for i in range(len(nodes)) if np.array_equal(nodes[i]["bounds"], b)]
    assert roots
    for root in roots:
        sub = nodes[root:root + len(want)]
        if len(sub) != len(want):
            continue
        ok = np.array_equal(sub["bounds"], want["bounds"])
        if not ok:
            continue
        for k in range(len(want)):
            for side in ("left", "right"):
                w, g = int(want[k][side]), int(sub[k][side])
                wk = w >> 28
                if wk == F.RT_REF_NODE:
                    assert g >> 28 == F.RT_REF_NODE and (g & 0x0FFFFFFF) - root == (w & 0x0FFFFFFF)
                elif wk in (F.RT_REF_SPHERE, F.RT_REF_QUAD, F.RT_REF_TRI):
                    assert g == w
                elif wk == F.RT_REF_MEDIUM:
                    assert g >> 28 == F.RT_REF_MEDIUM
                else:
                    assert g >> 28 in (F.RT_REF_LIST, F.RT_REF_XFORM)
        return len(want)
    raise AssertionError("no subtree of the flat node array matches the oracle tree")


def test_bvh_split_order_book1():
    world, lights, cam = scenes.book1_final(seed=0)
    desc = rt.SceneDesc.from_world(world, lights)
    bvh_obj = int(desc.arrays["children"][desc.arrays["objects"][desc.world]["a"]])
    n = _compare_bvh(desc, bvh_obj)
    nsph = len(desc.arrays["spheres"])
    assert 470 <= nsph <= 490 and n >= 255


def test_bvh_split_order_moving_and_ties():
    world, lights, cam = scenes.book1_final(seed=3, moving=True, checker_ground=True)
    desc = rt.SceneDesc.from_world(world, lights)
    _compare_bvh(desc, int(desc.arrays["children"][desc.arrays["objects"][desc.world]["a"]]))
    # many equal keys: a regular grid of identical spheres stresses the stable-sort requirement
    lst = rt.HittableList()
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    for i in range(7):
        for j in range(5):
            for k in range(3):
                lst.add(rt.Sphere(rt.Point3(i % 3, j % 2, k), 0.3, m))
    desc = rt.SceneDesc.from_world(rt.BvhNode(lst))
    assert _compare_bvh(desc, desc.world) > 50


def test_bvh_split_order_book2_and_smoke():
    world, lights, cam = scenes.book2_final(seed=0)
    desc = rt.SceneDesc.from_world(world, lights)
    objs = desc.arrays["objects"]
    bvhs = [i for i in range(len(objs)) if objs[i]["kind"] == F.RT_OBJ_BVH]
    assert len(bvhs) == 2
    assert _compare_bvh(desc, bvhs[0]) == 511 and _compare_bvh(desc, bvhs[1]) == 1023
    world, lights, cam = scenes.cornell_smoke()
    desc = rt.SceneDesc.from_world(world, lights)
    objs = desc.arrays["objects"]
    bvhs = [i for i in range(len(objs)) if objs[i]["kind"] == F.RT_OBJ_BVH]
    _compare_bvh(desc, bvhs[0])


def test_bvh_split_order_triangle_mesh():
    desc = rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(5000, seed=11), rt.Lambertian(rt.Color(.5, .5, .5))))
    assert _compare_bvh(desc, desc.world) == _node_count(5000)
    host = rt.Scene(desc, host_only=True)
    assert host.info().n_nodes == _node_count(5000)


def _node_count(n):
    return 1 if n <= 2 else 1 + _node_count(n // 2) + _node_count(n - n // 2)


def test_node_count_formula():
    import json
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    for n, want in kat["bvh_node_count"]["cases"]:
        assert _node_count(n) == want
    for n in (1, 2, 3, 5, 400, 1000):
        desc = rt.SceneDesc.from_world(rt.SphereSet(np.random.default_rng(n).uniform(-1, 1, (n, 3)), np.full(n, 0.01),
                                                    rt.Lambertian(rt.Color(.5, .5, .5))))
        assert rt.Scene(desc, host_only=True).info().n_nodes == _node_count(n)


def test_object_bounds_equal_oracle():
    for build in (scenes.cornell_box, scenes.cornell_smoke, lambda: scenes.book2_final(seed=1), scenes.quads,
                  scenes.test_triangle, lambda: scenes.book1_final(seed=2, moving=True)):
        world, lights, cam = build()
        desc = rt.SceneDesc.from_world(world, lights)
        host, orc = rt.Scene(desc, host_only=True), O.OracleScene(desc)
        for i in range(len(desc.arrays["objects"])):
            assert np.array_equal(host.object_bounds(i), orc.object_bounds(i)), (build, i)


def test_camera_initialize_equals_oracle():
    for build in (scenes.cornell_box, scenes.book1_final, scenes.for_output13, scenes.book2_final):
        cam = build()[2].to_struct()
        a, b = F.rt_camera_derived(), O.camera_initialize(cam)
        F.check(F.lib.rt_camera_initialize(C.byref(cam), C.byref(a)))
        assert bytes(a) == bytes(b)


def test_resolve_equals_oracle_and_ppm(tmp_path):
    cam = rt.Camera()
    cam.image_width, cam.samples_per_pixel = 16, 4
    g = np.random.default_rng(1)
    acc = g.uniform(-0.5, 6.0, (16, 16, 3))
    acc[0, 0] = [np.nan, np.inf, -np.inf]
    img = rt.Image(cam, acc)
    assert np.array_equal(img.rgb, O.resolve_rgb8(cam.to_struct(), acc))
    p = tmp_path / "x.ppm"
    img.write_ppm(str(p))
    lines = p.read_text().split("\n")
    assert lines[:3] == ["P3", "16 16", "255"] and len(lines) == 3 + 256 + 1
    assert lines[3] == " ".join(str(int(v)) for v in img.rgb[0, 0])


def _walk_packed_beside_wide(s, n_tris):
    """Walks the 48-byte nodes and the 64-byte nodes they are made from side by side, from the roots: every node must
    have the same children with the same 8-bit boxes.  Returns how often every triangle is a leaf, and the sum of the
    leaf boxes' half areas (f32 boxes)."""
    wide = s.wide_nodes()
    packed, overlap = s.packed_nodes()
    root, refs, only = s.fast_tree_info()
    assert len(packed) > n_tris // 4 and wide[root]["has_box8"] == 1
    assert len(packed) <= len(wide)  # dense: the collapse of the reference tree keeps padding nodes in the 64-byte array
    seen = np.zeros(n_tris, dtype=int)
    leaf_area = 0.0
    todo = [(root, 0)]
    kind = lambda r: int(r) >> 28
    while todo:
        w, p = todo.pop()
        wc = {tuple(wide[w]["box8"][j]): (int(wide[w]["child"][j]), j) for j in range(4) if wide[w]["child"][j] != F.RT_REF_NONE}
        pc = [(tuple(packed[p]["box8"][j]), int(packed[p]["child"][j])) for j in range(4) if packed[p]["child"][j] != F.RT_REF_NONE]
        assert len(wc) == len(pc), (w, p)
        kinds = [kind(r) for _, r in pc]
        assert kinds == sorted(kinds, key=lambda k: k != F.RT_REF_NODE)  # inner children first
        for box, ref in pc:
            assert box in wc, (w, p, box)
            if kind(ref) == F.RT_REF_NODE:
                assert kind(wc[box][0]) == F.RT_REF_NODE
                todo.append((wc[box][0] & 0x0FFFFFFF, ref & 0x0FFFFFFF))
            else:
                assert ref == wc[box][0] and kind(ref) == F.RT_REF_TRI
                seen[ref & 0x0FFFFFFF] += 1
                b32 = wide[w]["box32"][wc[box][1]]
                e = b32[1::2] - b32[0::2]
                leaf_area += e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
    assert seen.sum() == refs
    return seen, leaf_area, overlap, only


def test_packed_tree_is_the_wide_tree_relaid(monkeypatch):
    """The 48-byte nodes of a triangle mesh (children addressed from two base indices, inner children first, triangles
    in leaf order) describe the same tree as the 64-byte nodes they are made from.  With the collapse of the reference's
    tree every triangle of the mesh is a leaf exactly once; the traversal-only tree (meshes of 1024 triangles or more)
    may name a triangle from several leaves."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    n_tris = 3001
    tri = scenes.triangle_soup(n_tris, seed=17, edge=0.06)
    area = lambda e: e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]
    monkeypatch.setenv("MI355RT_FAST_TREE_MIN", "0")
    s = rt.Scene(rt.Mesh(tri, m), host_only=True)
    seen, leaf_area, overlap, only = _walk_packed_beside_wide(s, n_tris)
    assert not only and (seen == 1).all()
    # boxes a random line through the world box crosses: sum of triangle box areas / world box area
    ext = tri.max(axis=1) - tri.min(axis=1)
    wb = np.array(s.info().world_bounds[:]).reshape(3, 2)
    assert abs(overlap - area(ext).sum() / area(wb[:, 1] - wb[:, 0])) < 0.05 * overlap
    monkeypatch.delenv("MI355RT_FAST_TREE_MIN")
    s2 = rt.Scene(rt.Mesh(tri, m), host_only=True)
    seen2, leaf_area2, overlap2, only2 = _walk_packed_beside_wide(s2, n_tris)
    assert only2 and (seen2 >= 1).all() and seen2.sum() <= 2 * n_tris
    assert abs(overlap2 - leaf_area2 / area(wb[:, 1] - wb[:, 0])) < 0.01 * overlap2 and overlap2 < overlap
    lst = rt.HittableList()
    lst.add(rt.Sphere(rt.Point3(0, 0, 0), 1.0, m))
    lst.add(rt.Triangle(rt.Point3(0, 0, 2), rt.Point3(1, 0, 2), rt.Point3(0, 1, 2), m))
    assert len(rt.Scene(rt.BvhNode(lst), host_only=True).packed_nodes()[0]) == 0  # not all triangles: no packed tree


def test_traversal_only_tree_covers_every_point_of_every_triangle():
    """The traversal-only tree splits space as well as objects: a triangle that straddles a split plane is a leaf on
    both sides, each time with the box of its part.  Whatever point of a triangle a ray hits, the ray must be able to
    reach the triangle through boxes that contain the point: from the root there has to be a chain of child boxes
    (8-bit boxes of the packed nodes, and the f32 boxes of the wide nodes) that all contain the point and ends in a
    leaf naming that triangle.  Checked for corners, edge midpoints, centroids and random interior points of every
    triangle of a soup (overlapping triangles) and of a closed surface (shared edges)."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    g = np.random.default_rng(3)
    for tri in (scenes.triangle_soup(2500, seed=23, edge=0.08), scenes.icosphere_mesh(subdiv=3)):
        n_tris = len(tri)
        s = rt.Scene(rt.Mesh(tri, m), host_only=True)
        root, refs, only = s.fast_tree_info()
        assert only and refs >= n_tris
        packed, _ = s.packed_nodes()
        wide = s.wide_nodes()
        w = g.dirichlet([1.0, 1.0, 1.0], size=(n_tris, 4))
        bary = np.concatenate([np.tile(np.array([[1., 0, 0], [0, 1., 0], [0, 0, 1.], [.5, .5, 0], [0, .5, .5], [.5, 0, .5], [1 / 3, 1 / 3, 1 / 3]]), (n_tris, 1, 1)), w], axis=1)
        pts = np.einsum("nkv,nvc->nkc", bary, tri)  # (n_tris, 11, 3)
        for nodes, start in ((packed, 0), (wide, root)):
            boxes = "box8" if nodes is packed else "box32"
            child = nodes["child"]
            bx = nodes[boxes]
            for t in range(n_tris):
                for p in pts[t]:
                    todo, found = [start], False
                    while todo and not found:
                        i = todo.pop()
                        for j in range(4):
                            r = int(child[i][j])
                            if r == F.RT_REF_NONE:
                                continue
                            b = bx[i][j]
                            if not (b[0] <= p[0] <= b[1] and b[2] <= p[1] <= b[3] and b[4] <= p[2] <= b[5]):
                                continue
                            if (r >> 28) == F.RT_REF_NODE:
                                todo.append(r & 0x0FFFFFFF)
                            elif r == ((F.RT_REF_TRI << 28) | t):
                                found = True
                                break
                    assert found, (t, p)


def test_errors_are_reported_not_panicked():
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    with pytest.raises(rt.RtError) as e:  # bvh.rs:17 indexes an empty vector
        rt.Scene(rt.BvhNode(rt.HittableList()), host_only=True)
    assert e.value.code == F.RT_E_INVALID
    with pytest.raises(rt.RtError) as e:  # bvh.rs:90-93 unwrap on NaN
        lst = rt.HittableList()
        for x in (0.0, float("nan"), 2.0):
            lst.add(rt.Sphere(rt.Point3(x, 0, 0), 0.5, m))
        rt.Scene(rt.BvhNode(lst), host_only=True)
    assert e.value.code == F.RT_E_INVALID and "NaN" in e.value.message
    with pytest.raises(rt.RtError) as e:  # hittable_list.rs:79 random_int(0, -1)
        rt.Scene(rt.HittableList.with_object(rt.Sphere(rt.Point3(0, 0, 0), 1, m)), rt.HittableList(), host_only=True)
    assert e.value.code == F.RT_E_INVALID
    inner = rt.ConstantMedium.new_with_color(rt.Sphere(rt.Point3(0, 0, 0), 1, m), 1.0, rt.Color(1, 1, 1))
    two = rt.ConstantMedium.new_with_color(inner, 1.0, rt.Color(1, 1, 1))  # a medium as the boundary of a medium: accepted
    rt.Scene(two, host_only=True).close()
    img = rt.ImageTexture(np.zeros((2, 2, 3), dtype=np.uint8))  # a phase function that reads u,v: accepted, list frames cost 6 more stack entries
    plain = rt.HittableList.with_object(rt.ConstantMedium.new_with_color(rt.Sphere(rt.Point3(0, 0, 0), 1, m), 1.0, rt.Color(1, 1, 1)))
    stale = rt.HittableList.with_object(rt.ConstantMedium.new_with_texture(rt.Sphere(rt.Point3(0, 0, 0), 1, m), 1.0, img))
    s_plain, s_stale = rt.Scene(plain, host_only=True), rt.Scene(stale, host_only=True)
    assert s_stale.info().max_stack == s_plain.info().max_stack + 6
    with pytest.raises(rt.RtError) as e:  # three deep: the kernels resolve two levels
        rt.Scene(rt.ConstantMedium.new_with_color(two, 1.0, rt.Color(1, 1, 1)), host_only=True)
    assert e.value.code == F.RT_E_UNSUPPORTED
    s = rt.Scene(rt.HittableList.with_object(rt.Sphere(rt.Point3(0, 0, 0), 1, m)), host_only=True)
    with pytest.raises(rt.RtError) as e:  # no device data: computing must fail loudly, never fall back
        s.closest_hit(np.zeros((1, 3)), np.ones((1, 3)))
    assert e.value.code == F.RT_E_HIP


def test_obj_reader_fan_triangulation(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# unit quad\nv 0 0 0\nv 1 0 0\nv 1 1 0.1\nv 0 1 0\nf 1 2 3 4\nf -4/1/1 -3/2/2 -2/3/3\n")
    from ray_tracer_amd.obj_loader import load_obj_triangles
    v = load_obj_triangles(str(p))
    assert v.shape == (3, 3, 3)
    assert np.array_equal(v[0], [[0, 0, 0], [1, 0, 0], [1, 1, np.float32(0.1)]])
    assert np.array_equal(v[1], [[0, 0, 0], [1, 1, np.float32(0.1)], [0, 1, 0]])
    assert np.array_equal(v[2], v[0])
    mesh = rt.Mesh.from_obj(str(p), rt.Lambertian(rt.Color(.5, .5, .5)))
    assert rt.Scene(mesh, host_only=True).info().n_nodes == 3


def _check_wide_tree(wn, leaf_points):
    """Every child box (f32 and, when present, 8-bit) must contain what lies below it; every primitive
    must be reachable exactly once.  leaf_points(ref) -> (k, 3) points the leaf box has to contain."""
    seen = []

    def contains(outer, inner):
        return all(outer[2 * a] <= inner[2 * a] and outer[2 * a + 1] >= inner[2 * a + 1] for a in range(3))

    def walk(i):
        node = wn[i]
        lo = np.full(3, np.inf)
        hi = np.full(3, -np.inf)
        for j in range(4):
            ref = int(node["child"][j])
            if ref == F.RT_REF_NONE:
                continue
            b32 = node["box32"][j]
            if node["has_box8"]:
                assert contains(node["box8"][j], b32), (i, j)
            if ref >> 28 == F.RT_REF_NODE:
                clo, chi = walk(ref & 0x0FFFFFFF)
            else:
                seen.append(ref)
                pts = leaf_points(ref)
                clo, chi = pts.min(axis=0), pts.max(axis=0)
            inner = [clo[0], chi[0], clo[1], chi[1], clo[2], chi[2]]
            assert contains(b32, inner), (i, j, b32, inner)
            lo = np.minimum(lo, [b32[0], b32[2], b32[4]])
            hi = np.maximum(hi, [b32[1], b32[3], b32[5]])
        return lo, hi

    walk(0)
    return seen


def test_wide_nodes_contain_their_children(monkeypatch):
    """The 4-wide collapse of the reference's tree (a mesh of 1024 triangles or more is normally walked through the
    traversal-only tree, whose leaves may hold parts of triangles: test_traversal_only_tree_covers_every_point_of_every_triangle)."""
    monkeypatch.setenv("MI355RT_FAST_TREE_MIN", "0")
    tris = scenes.triangle_soup(3000, seed=5)
    desc = rt.SceneDesc.from_world(rt.Mesh(tris, rt.Lambertian(rt.Color(.5, .5, .5))))
    wn = rt.Scene(desc, host_only=True).wide_nodes()
    assert len(wn) > 500 and wn[0]["has_box8"] == 1  # similar-sized triangles: the 8-bit grid is accepted
    seen = _check_wide_tree(wn, lambda ref: tris[ref & 0x0FFFFFFF])
    assert sorted(seen) == [(F.RT_REF_TRI << 28) | i for i in range(3000)]
    # on average the 8-bit boxes are only slightly larger than the f32 ones
    valid = wn["child"] != F.RT_REF_NONE
    d32 = wn["box32"][valid][:, 1::2] - wn["box32"][valid][:, 0::2]
    d8 = wn["box8"][valid][:, 1::2] - wn["box8"][valid][:, 0::2]
    assert np.all(d8 >= d32) and d8.sum() / d32.sum() < 1.05


def test_wide_nodes_fall_back_to_f32_when_one_primitive_dwarfs_the_rest():
    world, lights, cam = scenes.book1_final(seed=0)  # ground sphere of radius 1000 among spheres of radius 0.2
    desc = rt.SceneDesc.from_world(world, lights)
    wn = rt.Scene(desc, host_only=True).wide_nodes()
    assert len(wn) > 100 and wn[0]["has_box8"] == 0
    sph = desc.arrays["spheres"]

    def pts(ref):
        s = sph[ref & 0x0FFFFFFF]
        c, r = np.array(s["center"]), s["radius"]
        return np.stack([c - r, c + r])

    seen = _check_wide_tree(wn, pts)
    assert sorted(seen) == [(F.RT_REF_SPHERE << 28) | i for i in range(len(sph))]
    # a world that is not one BVH over primitives has no wide nodes
    world, lights, cam = scenes.cornell_box()
    assert len(rt.Scene(rt.SceneDesc.from_world(world, lights), host_only=True).wide_nodes()) == 0


def test_wide_nodes_report_boxes_as_their_lists():
    """A box is a list of six consecutive quads; inside the 4-wide nodes it is entered through a reference that
    names the quads directly, and the introspection call reports the list it stands for."""
    rng = np.random.default_rng(3)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    boxes = rt.HittableList()
    corners = []
    for _ in range(50):
        a = rng.uniform(0.0, 10.0, 3)
        b = a + rng.uniform(0.2, 1.0, 3)
        corners.append((a, b))
        boxes.add(rt.box_new(rt.Point3(*a), rt.Point3(*b), m))
    desc = rt.SceneDesc.from_world(rt.HittableList.with_object(rt.BvhNode(boxes)))
    sc = rt.Scene(desc, host_only=True)
    wn = sc.wide_nodes()
    assert len(wn) > 10
    leaves = [int(r) for r in wn["child"].reshape(-1) if r != F.RT_REF_NONE and (int(r) >> 28) != F.RT_REF_NODE]
    assert len(leaves) == 50 and len(set(leaves)) == 50
    assert all((r >> 28) == F.RT_REF_LIST for r in leaves)
    # the root's child boxes together cover every box of the scene
    root = wn[0]
    valid = root["child"] != F.RT_REF_NONE
    lo = root["box32"][valid][:, 0::2].min(axis=0)
    hi = root["box32"][valid][:, 1::2].max(axis=0)
    assert np.all(lo <= np.min([c[0] for c in corners], axis=0)) and np.all(hi >= np.max([c[1] for c in corners], axis=0))


def test_reference_side_binding_in_integration_md_covers_the_header():
    """The Rust `extern "C"` block a maintainer of the reference would add (it cannot be compiled here: no Rust
    toolchain) must name every function of the header, and its #[repr(C)] structs must list the fields of the
    ctypes structs in the same order."""
    header = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    for source in ("INTEGRATION.md", os.path.join("rust", "mi355rt-sys", "src", "lib.rs")):
        _check_rust_binding(open(os.path.join(ROOT, source)).read(), header)


def _check_rust_binding(text, header):
    declared = set(re.findall(r"^(?:int|void|double|const char\*)\s+(rt_[a-z0-9_]+)\s*\(", header, re.M))
    rust_fns = set(re.findall(r"pub fn (rt_[a-z0-9_]+)\s*\(", text))
    assert rust_fns == declared, rust_fns ^ declared
    for name, body in re.findall(r"pub struct (rt_[a-z0-9_]+)\s*\{([^}]*)\}", text):
        if name == "rt_scene":
            continue
        rust_fields = re.findall(r"pub ([a-z0-9_]+)\s*:", body)
        c_fields = [f[0] for f in getattr(F, name)._fields_]
        assert rust_fields == c_fields, (name, rust_fields, c_fields)


def _instanced_leaf(m, x, nest):
    """Translate(RotateY(list)) whose list is `nest` levels of two-member lists (inner list first, a sphere after
    it): entering every level leaves a list continuation (two entries) on the traversal stack."""
    inner = rt.box_new(rt.Point3(0, 0, 0), rt.Point3(0.5, 0.5, 0.5), m)
    for _ in range(nest):
        outer = rt.HittableList.with_object(inner)
        outer.add(rt.Sphere(rt.Point3(0.25, 0.8, 0.25), 0.1, m))
        inner = outer
    return rt.Translate(rt.RotateY(inner, 15.0), rt.Vec3(float(x), 0.0, 0.0))


def test_stack_need_counts_the_leaves_of_wide_nodes():
    """The ordered traversal defers up to three siblings per 4-wide level and a leaf then adds its own entries
    (an instance its restore entry, a list its continuation).  max_stack must cover the sum; when the 4-wide form
    needs more than the 32 entries the kernels carry the scene keeps its binary nodes, and beyond that it is
    rejected instead of overrunning the LDS stack."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))

    def scene(n_leaves, nest):
        lst = rt.HittableList()
        for x in range(n_leaves):
            lst.add(_instanced_leaf(m, x, nest))
        return rt.HittableList.with_object(rt.BvhNode(lst))

    # 256 leaves: binary depth 8 = four 4-wide levels (3 deferred siblings each); leaf: restore entry + nest * 2
    info = rt.Scene(scene(256, 2), host_only=True).info()
    assert info.max_stack >= 3 * 4 + 1 + 2 * 2 + 1
    wide = rt.Scene(scene(256, 2), host_only=True).wide_nodes()
    assert len(wide) > 0
    # 8 levels of lists: 12 + 1 + 16 entries and more exceed 32 with 4-wide nodes, the binary form (8 + 1 + 16 ...) fits
    deep = rt.Scene(scene(256, 8), host_only=True)
    assert 8 + 1 + 2 * 8 + 1 <= deep.info().max_stack <= 32
    # 13 levels: no form fits
    with pytest.raises(F.RtError) as e:
        rt.Scene(scene(256, 13), host_only=True)
    assert e.value.code == F.RT_E_TOO_LARGE


def _write_obj(path, rng, n_u=14, n_v=9):
    """A closed lumpy surface as an OBJ file exercising what a real asset can hold: decimal coordinates that are not
    f32 values, quads and pentagon fans next to triangles, v / v/vt / v//vn / v/vt/vn corners, negative (relative)
    indices, two groups, comments, blank lines, CRLF line ends, normals and texture coordinates to be skipped."""
    lines = ["# written by the test suite", "o lumpy", ""]
    verts = []
    for j in range(n_v + 1):
        for i in range(n_u):
            th, ph = np.pi * j / n_v, 2 * np.pi * i / n_u
            r = 1.0 + 0.15 * np.sin(3 * th) * np.cos(2 * ph)
            verts.append((r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)))
    for v in verts:
        lines.append("v %.9f %.9f %.9f" % v)
    lines += ["vt 0.5 0.5", "vn 0 1 0", "g upper"]
    idx = lambda i, j: j * n_u + (i % n_u) + 1
    forms = ("%d", "%d/1", "%d//1", "%d/1/1")
    for j in range(n_v):
        if j == n_v // 2:
            lines.append("g lower\r")
        for i in range(n_u):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            f = forms[(i + j) % 4]
            if (i + j) % 3 == 0:   # a quad
                lines.append("f " + " ".join(f % k for k in (a, b, c, d)))
            elif (i + j) % 3 == 1:  # two triangles, the second with relative indices
                lines.append("f " + " ".join(f % k for k in (a, b, c)))
                lines.append("f " + " ".join(str(k - len(verts) - 1) for k in (a, c, d)))
            else:                   # a pentagon: the quad with one more corner on an edge (a repeated vertex keeps it planar enough)
                lines.append("f " + " ".join(f % k for k in (a, b, c, d, a)))
    open(path, "w").write("\n".join(lines) + "\n")
    return len(verts)


def test_obj_file_through_the_c_abi(tmp_path):
    """rt_obj_load (the library's reader, what Mesh::from_obj / ObjModel::load use here) against the independent
    Python reader: same triangles from the same file; ObjModel's scale and offset applied per component in f64
    (obj_loader.rs:51-67); errors reported, not thrown."""
    from ray_tracer_amd.obj_loader import load_obj, load_obj_triangles
    p = str(tmp_path / "lumpy.obj")
    _write_obj(p, np.random.default_rng(3))
    a, b = load_obj(p), load_obj_triangles(p)
    assert a.shape == b.shape and a.shape[0] > 250 and np.array_equal(a, b)
    assert np.array_equal(a, a.astype(np.float32).astype(np.float64))  # every coordinate is an f32 value (tobj stores f32)
    s = load_obj(p, 2.5, (10.0, -3.0, 0.125))
    assert np.array_equal(s, a * 2.5 + np.array([10.0, -3.0, 0.125]))
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    mesh, model = rt.Mesh.from_obj(p, m), rt.ObjModel.load(p, m, 2.5, rt.Point3(10.0, -3.0, 0.125))
    assert np.array_equal(mesh.vertices, a) and np.array_equal(model.vertices, s)
    # bounding_box() of the ObjModel: from_points(min, max) over the transformed vertices (obj_loader.rs:103-105)
    desc = rt.SceneDesc.from_world(model)
    want = O.OracleScene(desc).object_bounds(desc.world)
    assert np.array_equal(rt.Scene(desc, host_only=True).object_bounds(desc.world), want)
    with pytest.raises(F.RtError) as e:
        load_obj(str(tmp_path / "missing.obj"))
    assert e.value.code == F.RT_E_INVALID and "cannot open" in e.value.message
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(F.RtError) as e:
        load_obj(str(bad))
    assert e.value.code == F.RT_E_INVALID and "out of range" in e.value.message
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    assert load_obj(str(empty)).shape == (0, 3, 3)


def test_traversal_only_tree_does_not_depend_on_the_threads_that_build_it():
    """The traversal-only tree is built by tasks that run on whatever threads are free and take their node numbers in
    blocks: the records must come out the same every time, byte for byte (wide nodes, packed nodes, exact-cull tree
    behind them), large enough a mesh for the parallel paths (chunked binning from 2^17 references on, subtrees as tasks
    from 2^13 on)."""
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    tri = scenes.triangle_soup(150000, seed=29, edge=0.03)
    first = None
    for _ in range(3):
        s = rt.Scene(rt.Mesh(tri, m), host_only=True)
        packed, overlap = s.packed_nodes()
        got = (s.wide_nodes().tobytes(), packed.tobytes(), overlap, s.fast_tree_info())
        s.close()
        if first is None:
            first = got
            assert got[3][2] and got[3][1] > len(tri)  # a traversal-only tree with spatial splits
        else:
            assert got == first
