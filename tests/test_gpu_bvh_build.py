"""BvhNode::from_objects (bvh.rs:20-77) built on the device (bvh_build.hip: level-synchronous, two stable radix sorts
per level) against the oracle's own tree (its pre-order dump: bounds, children, which primitive sits where) and against
the host recursion of the flattener: the node array (f64 bounds, child references, flags) and the 4-wide nodes derived
from it must be identical byte for byte."""
import numpy as np
import pytest

import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F
from ray_tracer_amd import scenes

pytestmark = pytest.mark.gpu


def _same_tree(desc, monkeypatch, min_objects=2):
    monkeypatch.setenv("MI355RT_DEVICE_BVH_MIN", str(min_objects))
    dev = rt.Scene(desc)                         # BVHs of at least min_objects objects are built on the GPU
    host = rt.Scene(desc, host_only=True)        # always the host recursion
    a, b = dev.nodes(), host.nodes()
    assert a.shape == b.shape and len(a) > 0
    assert a.tobytes() == b.tobytes(), np.nonzero(a != b)[0][:10]
    wa, wb = dev.wide_nodes(), host.wide_nodes()
    assert wa.tobytes() == wb.tobytes()
    ia, ib = dev.info(), host.info()
    assert (ia.n_nodes, ia.max_stack, ia.world_ref) == (ib.n_nodes, ib.max_stack, ib.world_ref)
    dev.close()
    return len(a)


def test_device_built_bvh_equals_host_recursion(monkeypatch):
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    # every span size from 1 up: the node-count recurrence, spans of one / two / three, odd and even splits
    rng = np.random.default_rng(7)
    for n in list(range(1, 40)) + [63, 64, 65, 100, 127, 128, 129, 1000, 4097]:
        _same_tree(rt.SceneDesc.from_world(rt.SphereSet(rng.uniform(-5, 5, (n, 3)), rng.uniform(0.05, 0.4, n), m, bvh=True)), monkeypatch, 1)
    # equal keys: a regular grid (every split meets ties, on every axis), spheres and boxes of equal extent
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(9.0), np.arange(7.0), indexing="ij"), -1).reshape(-1, 3)
    _same_tree(rt.SceneDesc.from_world(rt.SphereSet(g, np.full(len(g), 0.25), m, bvh=True)), monkeypatch)
    _same_tree(rt.SceneDesc.from_world(rt.SphereSet(np.zeros((300, 3)), np.full(300, 1.0), m, bvh=True)), monkeypatch)  # all keys equal
    # signed zeros among the keys: -0.0 and +0.0 compare equal in the reference's comparator
    z = rng.uniform(-1, 1, (200, 3))
    z[::3, 0] = 0.0
    z[1::3, 0] = -0.0
    tri = np.stack([z, z + [0.0, 0.1, 0.0], z + [0.0, 0.0, 0.1]], axis=1)
    _same_tree(rt.SceneDesc.from_world(rt.Mesh(tri, m)), monkeypatch)
    # the scenes: Book-1 (BVH of spheres incl. the huge ground sphere), Book-2 (two BVHs, one under transforms), a mesh
    for build in (lambda: scenes.book1_final(seed=0, moving=True), lambda: scenes.book2_final(seed=0)):
        world, lights, cam = build()
        _same_tree(rt.SceneDesc.from_world(world, lights), monkeypatch)
    _same_tree(rt.SceneDesc.from_world(rt.Mesh(scenes.icosphere_mesh(subdiv=5), m)), monkeypatch)


def test_device_built_bvh_of_the_benchmark_mesh(monkeypatch):
    """2^20 triangles: the default threshold sends this one to the device; same bytes as the host build, and the
    closest hits through it equal the oracle's (tests/test_gpu_fullsize.py traces it at full batch size)."""
    desc = rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 20, seed=2))
    monkeypatch.delenv("MI355RT_DEVICE_BVH_MIN", raising=False)
    dev, host = rt.Scene(desc), rt.Scene(desc, host_only=True)
    a, b = dev.nodes(), host.nodes()
    assert len(a) == (1 << 20) - 1 and a.tobytes() == b.tobytes()
    assert dev.wide_nodes().tobytes() == host.wide_nodes().tobytes()


def test_device_built_bvh_equals_the_oracle_tree(monkeypatch):
    """The device build against BvhNode::from_objects as the oracle restates it, node for node in pre-order: split axis
    by longest extent (bvh.rs:27-33), stable order among equal keys (bvh.rs:44-46), spans of one, two and more objects."""
    from test_host_flatten import _compare_bvh
    monkeypatch.setenv("MI355RT_DEVICE_BVH_MIN", "1")
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    rng = np.random.default_rng(11)
    cases = []
    for n in (1, 2, 3, 5, 8, 33, 100, 1000):
        cases.append(rt.SceneDesc.from_world(rt.SphereSet(rng.uniform(-5, 5, (n, 3)), rng.uniform(0.05, 0.4, n), m, bvh=True)))
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(5.0), np.arange(4.0), indexing="ij"), -1).reshape(-1, 3)
    cases.append(rt.SceneDesc.from_world(rt.SphereSet(g, np.full(len(g), 0.25), m, bvh=True)))                       # ties on every axis
    cases.append(rt.SceneDesc.from_world(rt.Mesh(scenes.icosphere_mesh(subdiv=4), m)))
    cases.append(rt.SceneDesc.from_world(scenes.triangle_soup_scene(1 << 15, seed=2)))
    for desc in cases:
        dev = rt.Scene(desc)
        obj = desc.world
        o = desc.arrays["objects"][obj]
        if o["kind"] == F.RT_OBJ_LIST and o["b"] == 1 and o["child_kind"] == F.RT_CHILD_OBJECTS:
            obj = int(desc.arrays["children"][o["a"]])
        assert _compare_bvh(desc, obj, scene=dev) >= 1
        dev.close()
    world, lights, cam = scenes.book1_final(seed=0, moving=True)
    desc = rt.SceneDesc.from_world(world, lights)
    dev = rt.Scene(desc)
    assert _compare_bvh(desc, int(desc.arrays["children"][desc.arrays["objects"][desc.world]["a"]]), scene=dev) >= 255
    dev.close()
