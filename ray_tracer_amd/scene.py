"""Scene description builder and the handle on a flattened scene.

`SceneDesc.from_world(world, lights)` walks the host object graph (hittable.py) and emits
the POD arrays of `rt_scene_desc`; `Scene` hands them to `rt_scene_create`, which builds
every BVH in the reference's split order, flattens and uploads, and then answers batched
closest-hit queries and renders through the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

from . import _ffi as F
from .hittable import (BvhNode, ConstantMedium, Hittable, HittableList, Quad, RotateY, Sphere, SphereSet, Translate,
                       Triangle, TriangleSet)
from .material import Dielectric, DiffuseLight, Isotropic, Lambertian, Material, Metal
from .texture import CheckerTexture, ImageTexture, NoiseTexture, SolidColor, Texture

SPHERE_DT = np.dtype([("center", "<f8", 3), ("center2", "<f8", 3), ("radius", "<f8"), ("mat", "<i4"), ("moving", "<i4")])
QUAD_DT = np.dtype([("q", "<f8", 3), ("u", "<f8", 3), ("v", "<f8", 3), ("mat", "<i4"), ("_pad", "<i4")])
TRI_DT = np.dtype([("v0", "<f8", 3), ("v1", "<f8", 3), ("v2", "<f8", 3), ("mat", "<i4"), ("_pad", "<i4")])
OBJ_DT = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("child_kind", "<i4"), ("mat", "<i4"), ("_pad", "<i4"),
                   ("param", "<f8", 3)])
MAT_DT = np.dtype([("kind", "<i4"), ("tex", "<i4"), ("albedo", "<f8", 3), ("fuzz", "<f8"), ("ior", "<f8")])
TEX_DT = np.dtype([("kind", "<i4"), ("even", "<i4"), ("odd", "<i4"), ("image", "<i4"), ("perlin", "<i4"), ("_pad", "<i4"),
                   ("color", "<f8", 3), ("scale", "<f8")])
IMG_DT = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<u8")])
PERLIN_DT = np.dtype([("randvec", "<f8", (256, 3)), ("perm_x", "<i4", 256), ("perm_y", "<i4", 256), ("perm_z", "<i4", 256)])

assert SPHERE_DT.itemsize == C.sizeof(F.rt_sphere) == 64
assert QUAD_DT.itemsize == C.sizeof(F.rt_quad) == 80
assert TRI_DT.itemsize == C.sizeof(F.rt_triangle) == 80
assert OBJ_DT.itemsize == C.sizeof(F.rt_object) == 48
assert MAT_DT.itemsize == C.sizeof(F.rt_material) == 48
assert TEX_DT.itemsize == C.sizeof(F.rt_texture) == 56
assert IMG_DT.itemsize == C.sizeof(F.rt_image) == 16
assert PERLIN_DT.itemsize == C.sizeof(F.rt_perlin) == 9216


def _v(x) -> tuple:
    return (float(x[0]), float(x[1]), float(x[2]))


class SceneDesc:
    """POD arrays of one scene + the `rt_scene_desc` struct pointing at them."""

    def __init__(self):
        self._spheres: List[np.ndarray] = []
        self._quads: List[tuple] = []
        self._tris: List[np.ndarray] = []
        self._objects: List[tuple] = []
        self._children: List[int] = []
        self._materials: List[tuple] = []
        self._textures: List[tuple] = []
        self._images: List[tuple] = []
        self._image_bytes: List[np.ndarray] = []
        self._image_off = 0
        self._perlins: List[np.ndarray] = []
        self._n_spheres = 0
        self._n_tris = 0
        self._memo: Dict[int, int] = {}
        self._keep: List[object] = []  # keeps memoised host objects alive (ids must stay unique)
        self.world = -1
        self.lights = -1
        self.struct: Optional[F.rt_scene_desc] = None
        self.arrays: Dict[str, np.ndarray] = {}

    # ------------------------------------------------------------------ leaves
    def _texture(self, t: Texture) -> int:
        key = id(t)
        if key in self._memo:
            return self._memo[key]
        if isinstance(t, SolidColor):
            rec = (F.RT_TEX_SOLID, -1, -1, -1, -1, 0, _v(t.albedo), 0.0)
        elif isinstance(t, CheckerTexture):
            e, o = self._texture(t.even), self._texture(t.odd)
            rec = (F.RT_TEX_CHECKER, e, o, -1, -1, 0, (0.0, 0.0, 0.0), t.scale)
        elif isinstance(t, ImageTexture):
            img = -1
            if t.rtw_image.pixels is not None:
                px = t.rtw_image.pixels
                img = len(self._images)
                self._images.append((t.rtw_image.width, t.rtw_image.height, self._image_off))
                self._image_bytes.append(px.reshape(-1))
                self._image_off += px.size
            rec = (F.RT_TEX_IMAGE, -1, -1, img, -1, 0, (0.0, 0.0, 0.0), 0.0)
        elif isinstance(t, NoiseTexture):
            p = np.zeros((), dtype=PERLIN_DT)
            p["randvec"] = np.array([_v(v) for v in t.noise.randvec], dtype=np.float64)
            p["perm_x"], p["perm_y"], p["perm_z"] = t.noise.perm_x, t.noise.perm_y, t.noise.perm_z
            self._perlins.append(p)
            rec = (F.RT_TEX_NOISE, -1, -1, -1, len(self._perlins) - 1, 0, (0.0, 0.0, 0.0), t.scale)
        else:
            raise TypeError(f"unknown texture {type(t).__name__}")
        self._textures.append(rec)
        self._keep.append(t)
        self._memo[key] = len(self._textures) - 1
        return self._memo[key]

    def _material(self, m: Material) -> int:
        key = id(m)
        if key in self._memo:
            return self._memo[key]
        z = (0.0, 0.0, 0.0)
        if isinstance(m, Lambertian):
            rec = (F.RT_MAT_LAMBERTIAN, self._texture(m.tex), z, 0.0, 0.0)
        elif isinstance(m, Metal):
            rec = (F.RT_MAT_METAL, -1, _v(m.albedo), m.fuzz, 0.0)
        elif isinstance(m, Dielectric):
            rec = (F.RT_MAT_DIELECTRIC, -1, z, 0.0, m.refraction_index)
        elif isinstance(m, DiffuseLight):
            rec = (F.RT_MAT_DIFFUSE_LIGHT, self._texture(m.tex), z, 0.0, 0.0)
        elif isinstance(m, Isotropic):
            rec = (F.RT_MAT_ISOTROPIC, self._texture(m.tex), z, 0.0, 0.0)
        else:
            raise TypeError(f"unknown material {type(m).__name__}")
        self._materials.append(rec)
        self._keep.append(m)
        self._memo[key] = len(self._materials) - 1
        return self._memo[key]

    def _add_spheres(self, arr: np.ndarray) -> int:
        first = self._n_spheres
        self._spheres.append(arr)
        self._n_spheres += len(arr)
        return first

    def _add_tris(self, arr: np.ndarray) -> int:
        first = self._n_tris
        self._tris.append(arr)
        self._n_tris += len(arr)
        return first

    def _emit(self, kind, a=0, b=0, child_kind=0, mat=-1, param=(0.0, 0.0, 0.0)) -> int:
        self._objects.append((kind, a, b, child_kind, mat, 0, param))
        return len(self._objects) - 1

    # ------------------------------------------------------------------ graph
    def _object(self, h: Hittable) -> int:
        key = id(h)
        if key in self._memo:
            return self._memo[key]
        if isinstance(h, Sphere):
            s = np.zeros(1, dtype=SPHERE_DT)
            s["center"], s["center2"], s["radius"] = _v(h.center1), _v(h.center2), h.radius
            s["mat"], s["moving"] = self._material(h.mat), 1 if h.moving else 0
            oid = self._emit(F.RT_OBJ_SPHERE, self._add_spheres(s))
        elif isinstance(h, Quad):
            self._quads.append((_v(h.q), _v(h.u), _v(h.v), self._material(h.mat), 0))
            oid = self._emit(F.RT_OBJ_QUAD, len(self._quads) - 1)
        elif isinstance(h, Triangle):
            t = np.zeros(1, dtype=TRI_DT)
            t["v0"], t["v1"], t["v2"], t["mat"] = _v(h.v0), _v(h.v1), _v(h.v2), self._material(h.material)
            oid = self._emit(F.RT_OBJ_TRIANGLE, self._add_tris(t))
        elif isinstance(h, (HittableList, BvhNode)):
            ids = [self._object(c) for c in h.objects]
            first = len(self._children)
            self._children.extend(ids)
            kind = F.RT_OBJ_BVH if isinstance(h, BvhNode) else F.RT_OBJ_LIST
            oid = self._emit(kind, first, len(ids), F.RT_CHILD_OBJECTS)
        elif isinstance(h, Translate):
            oid = self._emit(F.RT_OBJ_TRANSLATE, self._object(h.object), param=_v(h.offset))
        elif isinstance(h, RotateY):
            oid = self._emit(F.RT_OBJ_ROTATE_Y, self._object(h.object), param=(h.angle, 0.0, 0.0))
        elif isinstance(h, ConstantMedium):
            oid = self._emit(F.RT_OBJ_CONSTANT_MEDIUM, self._object(h.boundary), mat=self._material(h.phase_function),
                             param=(h.density, 0.0, 0.0))
        elif isinstance(h, TriangleSet):
            t = np.zeros(len(h.vertices), dtype=TRI_DT)
            t["v0"], t["v1"], t["v2"] = h.vertices[:, 0], h.vertices[:, 1], h.vertices[:, 2]
            t["mat"] = self._material(h.material)
            first = self._add_tris(t)
            if h.bvh:
                oid = self._emit(F.RT_OBJ_BVH, first, len(t), F.RT_CHILD_TRIANGLES)
            else:
                oid = self._emit(F.RT_OBJ_OBJMODEL, first, len(t))
        elif isinstance(h, SphereSet):
            s = np.zeros(len(h.centers), dtype=SPHERE_DT)
            s["center"], s["center2"], s["radius"] = h.centers, h.centers, h.radii
            s["mat"] = self._material(h.material)
            first = self._add_spheres(s)
            oid = self._emit(F.RT_OBJ_BVH if h.bvh else F.RT_OBJ_LIST, first, len(s), F.RT_CHILD_SPHERES)
        else:
            raise TypeError(f"unknown hittable {type(h).__name__}")
        self._keep.append(h)
        self._memo[key] = oid
        return oid

    # ------------------------------------------------------------------ finish
    @staticmethod
    def from_world(world: Hittable, lights: Optional[Hittable] = None) -> "SceneDesc":
        d = SceneDesc()
        d.world = d._object(world)
        d.lights = d._object(lights) if lights is not None else -1
        d._finish()
        return d

    def _finish(self) -> None:
        def cat(parts, dt):
            return np.concatenate(parts) if parts else np.zeros(0, dtype=dt)

        a = self.arrays
        a["spheres"] = cat(self._spheres, SPHERE_DT)
        a["quads"] = np.array(self._quads, dtype=QUAD_DT) if self._quads else np.zeros(0, dtype=QUAD_DT)
        a["triangles"] = cat(self._tris, TRI_DT)
        a["objects"] = np.array(self._objects, dtype=OBJ_DT)
        a["children"] = np.array(self._children, dtype=np.int32)
        a["materials"] = np.array(self._materials, dtype=MAT_DT) if self._materials else np.zeros(0, dtype=MAT_DT)
        a["textures"] = np.array(self._textures, dtype=TEX_DT) if self._textures else np.zeros(0, dtype=TEX_DT)
        a["images"] = np.array(self._images, dtype=IMG_DT) if self._images else np.zeros(0, dtype=IMG_DT)
        a["image_bytes"] = cat(self._image_bytes, np.uint8)
        a["perlins"] = np.array(self._perlins, dtype=PERLIN_DT) if self._perlins else np.zeros(0, dtype=PERLIN_DT)
        s = F.rt_scene_desc()
        for name in ("spheres", "quads", "triangles", "objects", "children", "materials", "textures", "images",
                     "image_bytes", "perlins"):
            arr = np.ascontiguousarray(a[name])
            a[name] = arr
            setattr(s, name, arr.ctypes.data if arr.size else None)
            setattr(s, "n_" + name, len(arr))
        s.world, s.lights = self.world, self.lights
        self.struct = s
        self._spheres, self._tris, self._image_bytes = [], [], []


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


HIT_FIELDS = ("t", "prim", "u", "v", "front_face", "mat", "nx", "ny", "nz", "px", "py", "pz")
_HIT_DTYPES = {"prim": np.uint32, "front_face": np.uint8, "mat": np.int32}


def empty_hits(n: int) -> Dict[str, np.ndarray]:
    return {f: np.zeros(n, dtype=_HIT_DTYPES.get(f, np.float64)) for f in HIT_FIELDS}


class Scene:
    """A flattened scene resident in HBM (`rt_scene*`)."""

    def __init__(self, world: Hittable | SceneDesc, lights: Optional[Hittable] = None, host_only: bool = False):
        self.desc = world if isinstance(world, SceneDesc) else SceneDesc.from_world(world, lights)
        self._h = C.c_void_p()
        F.check(F.lib.rt_scene_create(C.byref(self.desc.struct), F.RT_SCENE_HOST_ONLY if host_only else 0, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            F.lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def info(self) -> F.rt_scene_info:
        i = F.rt_scene_info()
        F.check(F.lib.rt_scene_get_info(self._h, C.byref(i)))
        return i

    def nodes(self) -> np.ndarray:
        n = self.info().n_nodes
        buf = (F.rt_bvh_node * max(1, n))()
        F.check(F.lib.rt_scene_get_nodes(self._h, 0, n, buf))
        dt = np.dtype([("bounds", "<f8", 6), ("left", "<u4"), ("right", "<u4"), ("flags", "<u4"), ("_pad", "<u4")])
        return np.frombuffer(buf, dtype=dt, count=n).copy()

    def wide_nodes(self) -> np.ndarray:
        """4-wide nodes of the primitive-BVH kernel (empty when the world is not one BVH over primitives)."""
        cnt = C.c_uint64()
        F.check(F.lib.rt_scene_wide_node_count(self._h, C.byref(cnt)))
        n = cnt.value
        buf = (F.rt_wide_node * max(1, n))()
        F.check(F.lib.rt_scene_get_wide_nodes(self._h, 0, n, buf))
        dt = np.dtype([("child", "<u4", 4), ("box32", "<f8", (4, 6)), ("box8", "<f8", (4, 6)), ("has_box8", "<i4"), ("_pad", "<i4")])
        return np.frombuffer(buf, dtype=dt, count=n).copy()

    def packed_nodes(self):
        """(nodes, leaf_overlap): the packed 48-byte tree the primitive-BVH kernel reads for a triangle mesh, decoded
        like wide_nodes() (node 0 is the root; empty when the world is not one BVH over triangles)."""
        cnt, overlap = C.c_uint64(), C.c_double()
        F.check(F.lib.rt_scene_packed_node_count(self._h, C.byref(cnt), C.byref(overlap)))
        n = cnt.value
        buf = (F.rt_wide_node * max(1, n))()
        F.check(F.lib.rt_scene_get_packed_nodes(self._h, 0, n, buf))
        dt = np.dtype([("child", "<u4", 4), ("box32", "<f8", (4, 6)), ("box8", "<f8", (4, 6)), ("has_box8", "<i4"), ("_pad", "<i4")])
        return np.frombuffer(buf, dtype=dt, count=n).copy(), overlap.value

    def fast_tree_info(self):
        """(index of the nearest-first kernel's root in wide_nodes(), triangle references of its packed form, traversal-only?)"""
        root, refs, only = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        F.check(F.lib.rt_scene_fast_tree_info(self._h, C.byref(root), C.byref(refs), C.byref(only)))
        return int(root.value), int(refs.value), bool(only.value)

    def object_bounds(self, obj: int) -> np.ndarray:
        out = (C.c_double * 6)()
        F.check(F.lib.rt_scene_object_bounds(self._h, obj, out))
        return np.array(out[:])

    # -------------------------------------------------------------- closest hit
    def closest_hit(self, origins: np.ndarray, directions: np.ndarray, time: Optional[np.ndarray] = None,
                    tmin: Optional[np.ndarray] = None, tmax: Optional[np.ndarray] = None, seed: int = 0,
                    flags: int = 0) -> Dict[str, np.ndarray]:
        """Batched `world.hit(ray, [tmin, tmax], rec)` on host arrays (copied to the device and back)."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).T)
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).T)
        n = o.shape[1]
        opt = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (time, tmin, tmax)]
        rays = F.rt_ray_batch(_ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), *[_ptr(a) for a in opt])
        hits = empty_hits(n)
        hb = F.rt_hit_batch(*[_ptr(hits[f]) for f in HIT_FIELDS])
        F.check(F.lib.rt_closest_hit(self._h, C.byref(rays), n, C.byref(hb), seed, flags))
        return hits

    def closest_hit_device(self, rays: Dict[str, "object"], hits: Dict[str, "object"], n: int, seed: int = 0,
                           flags: int = 0, stream: int = 0, noalloc: bool = False) -> None:
        """Same query on arrays already resident in HBM: `rays` / `hits` map field names to objects
        with a `data_ptr()` (torch tensors); absent optional fields are passed as NULL.
        `noalloc` takes the entry point that allocates nothing (at most RT_NOALLOC_MAX_RAYS rays)."""
        rb = F.rt_ray_batch(*[rays[f].data_ptr() if f in rays else None
                              for f in ("ox", "oy", "oz", "dx", "dy", "dz", "time", "tmin", "tmax")])
        hb = F.rt_hit_batch(*[hits[f].data_ptr() if f in hits else None for f in HIT_FIELDS])
        fn = F.lib.rt_closest_hit_device_noalloc if noalloc else F.lib.rt_closest_hit_device
        F.check(fn(self._h, C.byref(rb), n, C.byref(hb), seed, flags, C.c_void_p(stream)))

    def reserve(self, max_rays: int = 0, max_paths: int = 0) -> None:
        """Sizes the scene's device scratch ahead of time, so that later *_device calls within these sizes only enqueue."""
        F.check(F.lib.rt_scene_reserve(self._h, max_rays, max_paths))

    # ------------------------------------------------------------------- render
    def render(self, cam: F.rt_camera, seed: int = 0, work: Optional[F.rt_work] = None,
               accum: Optional[np.ndarray] = None) -> np.ndarray:
        d = F.rt_camera_derived()
        F.check(F.lib.rt_camera_initialize(C.byref(cam), C.byref(d)))
        if accum is None:
            accum = np.zeros((d.image_height, cam.image_width, 3), dtype=np.float64)
        F.check(F.lib.rt_render(self._h, C.byref(cam), seed, C.byref(work) if work is not None else None, accum.ctypes.data))
        return accum

    def render_device(self, cam: F.rt_camera, accum, seed: int = 0, work: Optional[F.rt_work] = None, stream: int = 0) -> None:
        F.check(F.lib.rt_render_device(self._h, C.byref(cam), seed, C.byref(work) if work is not None else None,
                                       C.c_void_p(accum.data_ptr()), C.c_void_p(stream)))

    def counters(self) -> F.rt_counters:
        c = F.rt_counters()
        F.check(F.lib.rt_get_counters(self._h, C.byref(c)))
        return c

    def reset_counters(self) -> None:
        F.check(F.lib.rt_reset_counters(self._h))
