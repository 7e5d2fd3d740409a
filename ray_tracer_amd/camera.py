"""Host mirror of camera.rs: the public fields of `Camera` (camera.rs:15-27) and
`render(world, lights)` (camera.rs:78-158).  The per-pixel sampling loop runs as the
wavefront path tracer in libmi355rt.so; the host only resolves and writes the image."""
from __future__ import annotations

import ctypes as C
import sys
from typing import Optional

import numpy as np

from . import _ffi as F
from .hittable import Hittable
from .scene import Scene
from .vec3 import Color, Point3, Vec3


class Image:
    """Result of a render: per-pixel radiance sums and the resolved 8-bit picture."""

    def __init__(self, cam: "Camera", accum: np.ndarray):
        self.accum = accum
        self.height, self.width = accum.shape[0], accum.shape[1]
        rgb = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        c = cam.to_struct()
        F.check(F.lib.rt_resolve_rgb8(C.byref(c), accum.ctypes.data, rgb.ctypes.data))
        self.rgb = rgb

    def write_ppm(self, path: str) -> None:  # the P3 text of camera.rs:86,151-154
        F.check(F.lib.rt_write_ppm(path.encode(), self.width, self.height, self.rgb.ctypes.data))


class Camera:
    def __init__(self):  # defaults of Camera::new (camera.rs:47-74)
        self.aspect_ratio = 1.0
        self.image_width = 100
        self.samples_per_pixel = 10
        self.max_depth = 10
        self.russian_roulette = False  # the termination rule of the reference's history (camera.rs:407-416); off at HEAD
        self.background = Color(0.0, 0.0, 0.0)
        self.vfov = 90.0
        self.lookfrom = Point3(0.0, 0.0, 0.0)
        self.lookat = Point3(0.0, 0.0, -1.0)
        self.vup = Vec3(0.0, 1.0, 0.0)
        self.defocus_angle = 0.0
        self.focus_dist = 10.0

    def to_struct(self) -> F.rt_camera:
        c = F.rt_camera()
        c.aspect_ratio, c.image_width = float(self.aspect_ratio), int(self.image_width)
        c.samples_per_pixel, c.max_depth = int(self.samples_per_pixel), int(self.max_depth)
        c.russian_roulette = 1 if self.russian_roulette else 0
        c.background = (C.c_double * 3)(*self.background)
        c.vfov = float(self.vfov)
        c.lookfrom = (C.c_double * 3)(*self.lookfrom)
        c.lookat = (C.c_double * 3)(*self.lookat)
        c.vup = (C.c_double * 3)(*self.vup)
        c.defocus_angle, c.focus_dist = float(self.defocus_angle), float(self.focus_dist)
        return c

    def derived(self) -> F.rt_camera_derived:  # Camera::initialize (camera.rs:161-205)
        d = F.rt_camera_derived()
        c = self.to_struct()
        F.check(F.lib.rt_camera_initialize(C.byref(c), C.byref(d)))
        return d

    def render(self, world: Hittable | Scene, lights: Optional[Hittable] = None, seed: int = 0,
               out: Optional[str] = None) -> Image:
        """camera.rs:78-158.  `lights=None` renders with the material PDF alone (the reference
        at HEAD requires a light list; its earlier revisions did not have one)."""
        scene = world if isinstance(world, Scene) else Scene(world, lights)
        accum = scene.render(self.to_struct(), seed=seed)
        img = Image(self, accum)
        if out is not None:
            img.write_ppm(out)
            print("Done.", file=sys.stderr)  # camera.rs:156
        return img
