"""ctypes binding of libmi355rt.so (the C ABI declared in include/mi355rt.h).

The library is the product: importing this module fails loudly if the HIP shared
object has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
There is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355RT_LIB: load a measurement build of the same library instead (ablations, phase timers); never a fallback
LIB_PATH = os.environ.get("MI355RT_LIB") or os.path.join(_HERE, "libmi355rt.so")

RT_OK = 0
RT_E_INVALID = -1
RT_E_HIP = -2
RT_E_UNSUPPORTED = -3
RT_E_TOO_LARGE = -4
RT_E_BUSY = -5

RT_SCENE_HOST_ONLY = 1

RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE = 0, 1, 2, 3
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = 0, 1, 2, 3, 4
(RT_OBJ_SPHERE, RT_OBJ_QUAD, RT_OBJ_TRIANGLE, RT_OBJ_LIST, RT_OBJ_BVH, RT_OBJ_TRANSLATE,
 RT_OBJ_ROTATE_Y, RT_OBJ_CONSTANT_MEDIUM, RT_OBJ_OBJMODEL) = range(9)
RT_CHILD_OBJECTS, RT_CHILD_SPHERES, RT_CHILD_QUADS, RT_CHILD_TRIANGLES = 0, 1, 2, 3

RT_REF_NODE, RT_REF_SPHERE, RT_REF_QUAD, RT_REF_TRI, RT_REF_LIST, RT_REF_XFORM, RT_REF_MEDIUM = range(7)
RT_REF_NONE = 0xFFFFFFFF

RT_HIT_REFERENCE_ORDER = 0
RT_HIT_ORDERED = 1
RT_HIT_COUNTERS = 2
RT_HIT_EXACT_CULL = 4
RT_NOALLOC_MAX_RAYS = 1 << 24

_d3 = C.c_double * 3


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("image", C.c_int32),
                ("perlin", C.c_int32), ("_pad", C.c_int32), ("color", _d3), ("scale", C.c_double)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("offset", C.c_uint64)]


class rt_perlin(C.Structure):
    _fields_ = [("randvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256),
                ("perm_y", C.c_int32 * 256), ("perm_z", C.c_int32 * 256)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tex", C.c_int32), ("albedo", _d3), ("fuzz", C.c_double), ("ior", C.c_double)]


class rt_sphere(C.Structure):
    _fields_ = [("center", _d3), ("center2", _d3), ("radius", C.c_double), ("mat", C.c_int32), ("moving", C.c_int32)]


class rt_quad(C.Structure):
    _fields_ = [("q", _d3), ("u", _d3), ("v", _d3), ("mat", C.c_int32), ("_pad", C.c_int32)]


class rt_triangle(C.Structure):
    _fields_ = [("v0", _d3), ("v1", _d3), ("v2", _d3), ("mat", C.c_int32), ("_pad", C.c_int32)]


class rt_object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("child_kind", C.c_int32),
                ("mat", C.c_int32), ("_pad", C.c_int32), ("param", _d3)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
        ("quads", C.c_void_p), ("n_quads", C.c_uint64),
        ("triangles", C.c_void_p), ("n_triangles", C.c_uint64),
        ("objects", C.c_void_p), ("n_objects", C.c_uint64),
        ("children", C.c_void_p), ("n_children", C.c_uint64),
        ("materials", C.c_void_p), ("n_materials", C.c_uint64),
        ("textures", C.c_void_p), ("n_textures", C.c_uint64),
        ("images", C.c_void_p), ("n_images", C.c_uint64),
        ("image_bytes", C.c_void_p), ("n_image_bytes", C.c_uint64),
        ("perlins", C.c_void_p), ("n_perlins", C.c_uint64),
        ("world", C.c_int32), ("lights", C.c_int32),
    ]


class rt_bvh_node(C.Structure):
    _fields_ = [("bounds", C.c_double * 6), ("left", C.c_uint32), ("right", C.c_uint32),
                ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class rt_wide_node(C.Structure):
    _fields_ = [("child", C.c_uint32 * 4), ("box32", (C.c_double * 6) * 4), ("box8", (C.c_double * 6) * 4),
                ("has_box8", C.c_int32), ("_pad", C.c_int32)]


class rt_scene_info(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_lists", C.c_uint64), ("n_xforms", C.c_uint64), ("n_media", C.c_uint64),
                ("n_lights", C.c_uint64), ("world_ref", C.c_uint32), ("max_stack", C.c_uint32),
                ("device_bytes", C.c_uint64), ("world_bounds", C.c_double * 6)]


class rt_ray_batch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "time", "tmin", "tmax")]


class rt_hit_batch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "prim", "u", "v", "front_face", "mat", "nx", "ny", "nz", "px", "py", "pz")]


class rt_camera(C.Structure):
    _fields_ = [("aspect_ratio", C.c_double), ("image_width", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("russian_roulette", C.c_int32), ("background", _d3), ("vfov", C.c_double),
                ("lookfrom", _d3), ("lookat", _d3), ("vup", _d3), ("defocus_angle", C.c_double),
                ("focus_dist", C.c_double)]


class rt_camera_derived(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("sqrt_spp", C.c_int32), ("pixel_samples_scale", C.c_double),
                ("recip_sqrt_spp", C.c_double), ("center", _d3), ("pixel00_loc", _d3), ("pixel_delta_u", _d3),
                ("pixel_delta_v", _d3), ("defocus_disk_u", _d3), ("defocus_disk_v", _d3)]


class rt_work(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_end", C.c_int32)]


class rt_counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("paths", C.c_uint64),
                ("render_rays", C.c_uint64), ("extend_launches", C.c_uint64), ("ms_last_closest_hit", C.c_double),
                ("ms_last_render", C.c_double), ("ms_last_traversal", C.c_double), ("ms_last_reorder", C.c_double),
                ("ms_sum_closest_hit", C.c_double), ("ms_sum_traversal", C.c_double), ("ms_sum_reorder", C.c_double),
                ("timed_calls", C.c_uint64), ("shader_khz", C.c_uint64)]


# Every symbol include/mi355rt.h declares, with its prototype.
PROTOTYPES = {
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (C.c_int, [C.c_void_p]),
    "rt_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(rt_scene_info)]),
    "rt_scene_get_nodes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(rt_bvh_node)]),
    "rt_scene_wide_node_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rt_scene_get_wide_nodes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(rt_wide_node)]),
    "rt_scene_packed_node_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "rt_scene_fast_tree_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "rt_scene_get_packed_nodes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(rt_wide_node)]),
    "rt_scene_object_bounds": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "rt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.POINTER(rt_triangle)), C.POINTER(C.c_uint64)]),
    "rt_obj_free": (None, [C.POINTER(rt_triangle)]),
    "rt_scene_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "rt_closest_hit": (C.c_int, [C.c_void_p, C.POINTER(rt_ray_batch), C.c_uint64, C.POINTER(rt_hit_batch), C.c_uint64, C.c_uint32]),
    "rt_closest_hit_device": (C.c_int, [C.c_void_p, C.POINTER(rt_ray_batch), C.c_uint64, C.POINTER(rt_hit_batch), C.c_uint64, C.c_uint32, C.c_void_p]),
    "rt_closest_hit_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(rt_ray_batch), C.c_uint64, C.POINTER(rt_hit_batch), C.c_uint64, C.c_uint32, C.c_void_p]),
    "rt_camera_initialize": (C.c_int, [C.POINTER(rt_camera), C.POINTER(rt_camera_derived)]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.c_uint64, C.POINTER(rt_work), C.c_void_p]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.c_uint64, C.POINTER(rt_work), C.c_void_p, C.c_void_p]),
    "rt_resolve_rgb8": (C.c_int, [C.POINTER(rt_camera), C.c_void_p, C.c_void_p]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rt_get_counters": (C.c_int, [C.c_void_p, C.POINTER(rt_counters)]),
    "rt_reset_counters": (C.c_int, [C.c_void_p]),
    "rt_rng_uniform": (C.c_double, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
}


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mi355rt error {code}: {message}")
        self.code = code
        self.message = message


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels carry their own libamdhip64.so; if this library pulled in the
    system's copy first and torch initialised afterwards, the process would hold two runtimes (observed: torch then
    reports hipErrorNoDevice).  When torch is installed its copy is loaded first, without importing torch, and
    libmi355rt.so binds to it through the shared soname."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C ray_tracer_amd/csrc`). "
            "ray_tracer_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != RT_OK:
        raise RtError(rc, (lib.rt_last_error() or b"").decode("utf-8", "replace"))
