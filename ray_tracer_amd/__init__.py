"""ray_tracer_amd: the hot path of the lxyWS/ray_tracer path tracer (BVH traversal +
closest-hit intersection driven by the camera's sampling loop) as HIP kernels for
MI355X behind a C ABI (include/mi355rt.h), plus a host mirror of the reference's
Hittable / BvhNode / Camera::render surface that describes scenes for it."""
from ._ffi import RtError, lib  # noqa: F401  (fails loudly when libmi355rt.so is missing)
from .camera import Camera, Image  # noqa: F401
from .hittable import (BvhNode, ConstantMedium, Hittable, HittableList, Mesh, ObjModel, Quad, RotateY, Sphere,  # noqa: F401
                       SphereSet, Translate, Triangle, TriangleSet, box_new)
from .material import Dielectric, DiffuseLight, Isotropic, Lambertian, Material, Metal  # noqa: F401
from .scene import Scene, SceneDesc  # noqa: F401
from .texture import CheckerTexture, ImageTexture, NoiseTexture, Perlin, RtwImage, SolidColor, Texture  # noqa: F401
from .vec3 import Color, Point3, Vec3, cross, dot, unit_vector  # noqa: F401
from . import parallel  # noqa: F401,E402
