"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference.
Checker only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ray_tracer_amd import _ffi as F
from ray_tracer_amd.scene import HIT_FIELDS, SceneDesc, _ptr, empty_hits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} missing: build it with `make -C oracle` (or __graft_entry__.build())")

olib = C.CDLL(LIB_PATH)
olib.oracle_last_error.restype = C.c_char_p
olib.oracle_scene_create.restype = C.c_void_p
olib.oracle_scene_create.argtypes = [C.POINTER(F.rt_scene_desc)]
olib.oracle_scene_destroy.argtypes = [C.c_void_p]
olib.oracle_object_bounds.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
olib.oracle_bvh_dump.restype = C.c_int64
olib.oracle_bvh_dump.argtypes = [C.c_void_p, C.c_int32, C.POINTER(F.rt_bvh_node), C.c_uint64]
olib.oracle_closest_hit.argtypes = [C.c_void_p, C.POINTER(F.rt_ray_batch), C.c_uint64, C.POINTER(F.rt_hit_batch),
                                    C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
olib.oracle_camera_initialize.argtypes = [C.POINTER(F.rt_camera), C.POINTER(F.rt_camera_derived)]
olib.oracle_render.argtypes = [C.c_void_p, C.POINTER(F.rt_camera), C.c_uint64, C.POINTER(F.rt_work), C.c_void_p, C.c_int,
                               C.POINTER(C.c_uint64)]
olib.oracle_resolve_rgb8.argtypes = [C.POINTER(F.rt_camera), C.c_void_p, C.c_void_p]
olib.oracle_rng_uniform.restype = C.c_double
olib.oracle_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
olib.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
olib.oracle_quad_pdf_value.restype = C.c_double
olib.oracle_quad_pdf_value.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
olib.oracle_sphere_pdf_value.restype = C.c_double
olib.oracle_sphere_pdf_value.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
olib.oracle_texture_value.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
olib.oracle_aabb_hit.restype = C.c_int
olib.oracle_aabb_hit.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]

NODE_DT = np.dtype([("bounds", "<f8", 6), ("left", "<u4"), ("right", "<u4"), ("flags", "<u4"), ("_pad", "<u4")])


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleScene:
    def __init__(self, desc_or_world, lights=None):
        self.desc = desc_or_world if isinstance(desc_or_world, SceneDesc) else SceneDesc.from_world(desc_or_world, lights)
        self._h = olib.oracle_scene_create(C.byref(self.desc.struct))
        if not self._h:
            raise RuntimeError("oracle: " + olib.oracle_last_error().decode())

    def close(self):
        if self._h:
            olib.oracle_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def object_bounds(self, obj: int) -> np.ndarray:
        out = (C.c_double * 6)()
        assert olib.oracle_object_bounds(self._h, obj, out) == 0
        return np.array(out[:])

    def bvh_dump(self, obj: int) -> np.ndarray:
        n = olib.oracle_bvh_dump(self._h, obj, None, 0)
        assert n >= 0
        buf = (F.rt_bvh_node * max(1, n))()
        olib.oracle_bvh_dump(self._h, obj, buf, n)
        return np.frombuffer(buf, dtype=NODE_DT, count=n).copy()

    def closest_hit(self, origins, directions, time=None, tmin=None, tmax=None, seed=0, nthreads=8, counters=None):
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).T)
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).T)
        n = o.shape[1]
        opt = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (time, tmin, tmax)]
        rays = F.rt_ray_batch(_ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), *[_ptr(a) for a in opt])
        hits = empty_hits(n)
        hb = F.rt_hit_batch(*[_ptr(hits[f]) for f in HIT_FIELDS])
        ctr = (C.c_uint64 * 2)(0, 0)
        rc = olib.oracle_closest_hit(self._h, C.byref(rays), n, C.byref(hb), seed, nthreads, ctr)
        assert rc == 0
        if counters is not None:
            counters["node_visits"] = counters.get("node_visits", 0) + ctr[0]
            counters["prim_tests"] = counters.get("prim_tests", 0) + ctr[1]
        return hits

    def render(self, cam: F.rt_camera, seed=0, work=None, accum=None, nthreads=8, stats=None):
        d = F.rt_camera_derived()
        olib.oracle_camera_initialize(C.byref(cam), C.byref(d))
        if accum is None:
            accum = np.zeros((d.image_height, cam.image_width, 3), dtype=np.float64)
        nr = C.c_uint64(0)
        rc = olib.oracle_render(self._h, C.byref(cam), seed, C.byref(work) if work is not None else None,
                                accum.ctypes.data, nthreads, C.byref(nr))
        if rc != 0:
            raise RuntimeError("oracle: " + olib.oracle_last_error().decode())
        if stats is not None:
            stats["rays"] = stats.get("rays", 0) + nr.value
        return accum

    def quad_pdf_value(self, quad, origin, direction):
        return olib.oracle_quad_pdf_value(self._h, quad, _d3(origin), _d3(direction))

    def sphere_pdf_value(self, sphere, origin, direction):
        return olib.oracle_sphere_pdf_value(self._h, sphere, _d3(origin), _d3(direction))

    def texture_value(self, tex, u, v, p):
        out = (C.c_double * 3)()
        olib.oracle_texture_value(self._h, tex, u, v, _d3(p), out)
        return np.array(out[:])


def resolve_rgb8(cam: F.rt_camera, accum: np.ndarray) -> np.ndarray:
    rgb = np.zeros(accum.shape, dtype=np.uint8)
    olib.oracle_resolve_rgb8(C.byref(cam), accum.ctypes.data, rgb.ctypes.data)
    return rgb


def camera_initialize(cam: F.rt_camera) -> F.rt_camera_derived:
    d = F.rt_camera_derived()
    olib.oracle_camera_initialize(C.byref(cam), C.byref(d))
    return d


def aabb_hit(bounds, o, d, tmin, tmax) -> bool:
    return bool(olib.oracle_aabb_hit((C.c_double * 6)(*bounds), _d3(o), _d3(d), tmin, tmax))


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    olib.oracle_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)
