"""rt_closest_hit_device_noalloc (the entry point that allocates nothing: its per-ray scratch is a fixed
array inside the library's code object) against rt_closest_hit_device, through the C ABI, on device
buffers.  Bar: every output array identical bit for bit.  Two calls in a row on each scene, the second
with a smaller batch, so that stale scratch contents of the first call would show.

Scenes: the Cornell box (transform chains: the scratch holds non-zero contexts), the Book-2 final scene
(constant media, a hoisted first member: the split traversal), and a triangle mesh (the primitive-BVH
kernel, which needs no scratch), in both traversal orders.  The mesh batch is large enough (>= 2^18 rays on a tree
whose rays cross many leaf boxes) for the allocating entry point to run its binning pass, which the allocation-free
one must do without.  Free device memory is read before and after the allocation-free calls: they may not change it."""
import numpy as np
import pytest

import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F
from ray_tracer_amd import scenes
from ray_tracer_amd.scene import HIT_FIELDS

pytestmark = pytest.mark.gpu

_TORCH_DT = {"prim": "int32", "front_face": "uint8", "mat": "int32"}


def _device_rays(torch, dev, o, d, tm):
    rays = {}
    for k, nm in enumerate(("ox", "oy", "oz")):
        rays[nm] = torch.from_numpy(np.ascontiguousarray(o[:, k])).to(dev)
    for k, nm in enumerate(("dx", "dy", "dz")):
        rays[nm] = torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev)
    rays["time"] = torch.from_numpy(np.ascontiguousarray(tm)).to(dev)
    return rays


def _query(torch, dev, gpu, rays, n, flags, noalloc):
    hits = {f: torch.full((n,), 7, dtype=getattr(torch, _TORCH_DT.get(f, "float64")), device=dev) for f in HIT_FIELDS}
    r = {k: v[:n].contiguous() for k, v in rays.items()}
    gpu.closest_hit_device(r, hits, n, seed=5, flags=flags, stream=torch.cuda.current_stream().cuda_stream, noalloc=noalloc)
    torch.cuda.synchronize()
    return {f: hits[f].cpu().numpy() for f in HIT_FIELDS}


def _same(a, b):
    for f in HIT_FIELDS:
        x, y = a[f], b[f]
        if x.dtype == np.float64:
            x, y = x.view(np.uint64), y.view(np.uint64)  # bit for bit, NaN included
        assert np.array_equal(x, y), f


def test_noalloc_entry_point_matches_allocating_one():
    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    cases = []
    world, lights, _ = scenes.cornell_box()
    cases.append((rt.SceneDesc.from_world(world, lights), [1, 554, 1, 554, 1, 554]))
    world, lights, _ = scenes.book2_final(seed=0)
    cases.append((rt.SceneDesc.from_world(world, lights), [-200, 600, 0, 560, -200, 600]))
    cases.append((rt.SceneDesc.from_world(rt.Mesh(scenes.triangle_soup(20000, seed=13, edge=0.05), m)), [0, 1, 0, 1, 0, 1]))
    for case, (desc, box) in enumerate(cases):
        n_first, n_second = (300000, 33333) if case == 2 else (100000, 33333)
        gpu = rt.Scene(desc)
        o, d, tm = scenes.incoherent_rays(n_first, box, seed=41)
        rays = _device_rays(torch, dev, o, d, tm)
        for flags in (F.RT_HIT_REFERENCE_ORDER, F.RT_HIT_ORDERED):
            want_big = _query(torch, dev, gpu, rays, n_first, flags, noalloc=False)
            want_small = _query(torch, dev, gpu, rays, n_second, flags, noalloc=False)
            assert (want_big["prim"] != F.RT_REF_NONE).sum() > n_first // 10
            hits = {f: torch.empty((n_first,), dtype=getattr(torch, _TORCH_DT.get(f, "float64")), device=dev) for f in HIT_FIELDS}
            del hits  # (the caching allocator now holds the output buffers the calls below ask for)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(dev)[0]
            got_big = _query(torch, dev, gpu, rays, n_first, flags, noalloc=True)      # first call
            got_small = _query(torch, dev, gpu, rays, n_second, flags, noalloc=True)   # second call, smaller input
            assert torch.cuda.mem_get_info(dev)[0] == free_before, "the allocation-free entry point changed the free device memory"
            _same(got_big, want_big)
            _same(got_small, want_small)
        gpu.close()


def test_noalloc_entry_point_rejects_oversized_batches():
    import torch
    assert torch.cuda.is_available()
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    gpu = rt.Scene(rt.HittableList.with_object(rt.Sphere(rt.Point3(0, 0, -1), 0.5, m)))
    one = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    rb = F.rt_ray_batch(*([one.data_ptr()] * 6 + [None] * 3))
    prim = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    hb = F.rt_hit_batch(one.data_ptr(), prim.data_ptr(), *([None] * 10))
    import ctypes as C
    rc = F.lib.rt_closest_hit_device_noalloc(gpu.handle, C.byref(rb), F.RT_NOALLOC_MAX_RAYS + 1, C.byref(hb), 0, 0, None)
    assert rc == F.RT_E_TOO_LARGE and b"2^24" in F.lib.rt_last_error()
    gpu.close()


def test_overlapping_allocation_free_calls_are_refused():
    """All allocation-free calls on one GPU share one scratch array.  A call on another stream while the previous one's
    work is still running would overwrite it: the library returns RT_E_BUSY and enqueues nothing; once that work has
    finished, and always on the same stream, calls are accepted, from any scene."""
    import torch
    dev = torch.device("cuda", 0)
    world, lights, _ = scenes.cornell_box()
    gpu = rt.Scene(rt.SceneDesc.from_world(world, lights))
    world2, lights2, _ = scenes.book2_final(seed=0)
    other = rt.Scene(rt.SceneDesc.from_world(world2, lights2))
    n = 1 << 22
    o, d, tm = scenes.incoherent_rays(n, [1, 554, 1, 554, 1, 554], seed=43)
    rays = _device_rays(torch, dev, o, d, tm)
    hits = {f: torch.empty((n,), dtype=getattr(torch, _TORCH_DT.get(f, "float64")), device=dev) for f in HIT_FIELDS}
    hits2 = {f: torch.empty((1000,), dtype=getattr(torch, _TORCH_DT.get(f, "float64")), device=dev) for f in HIT_FIELDS}
    small = {k: v[:1000].contiguous() for k, v in rays.items()}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for scene in (gpu, other):  # (a kernel's first launch in a process loads its code and blocks the host for that long)
        scene.closest_hit_device(small, hits2, 1000, seed=5, flags=0, stream=s1.cuda_stream, noalloc=True)
    torch.cuda.synchronize()
    # some tens of milliseconds of other work in front of the call on its stream, so that the call's own work (about half
    # a millisecond) is certainly still pending when the next call arrives, however slow this host is
    a = torch.randn(4096, 4096, device=dev)
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        for _ in range(40):
            a = (a @ a).clamp_(-1.0, 1.0)
    gpu.closest_hit_device(rays, hits, n, seed=5, flags=0, stream=s1.cuda_stream, noalloc=True)
    with pytest.raises(rt.RtError) as err:
        other.closest_hit_device(small, hits2, 1000, seed=5, flags=0, stream=s2.cuda_stream, noalloc=True)
    assert err.value.code == F.RT_E_BUSY
    gpu.closest_hit_device(small, hits2, 1000, seed=5, flags=0, stream=s1.cuda_stream, noalloc=True)     # same stream: in order
    torch.cuda.synchronize()
    want = {f: hits2[f].cpu().numpy().copy() for f in HIT_FIELDS}
    gpu.closest_hit_device(small, hits2, 1000, seed=5, flags=0, stream=s2.cuda_stream, noalloc=True)     # the device is idle: accepted
    torch.cuda.synchronize()
    _same({f: hits2[f].cpu().numpy() for f in HIT_FIELDS}, want)
    gpu.close()
    other.close()
