"""Host mirror of the reference's Hittable objects (hittable.rs, hittable_list.rs, bvh.rs,
sphere.rs, quad.rs, triangle.rs, mesh.rs, obj_loader.rs, constant_medium.rs).

The classes keep the reference's names and constructor arguments; they only DESCRIBE
the object graph.  `Hittable.hit` of the reference becomes a batched query on the
flattened scene (ray_tracer_amd.scene.Scene.closest_hit), which runs the HIP kernels.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import numpy as np

from .material import Isotropic, Material
from .texture import Texture
from .vec3 import Color, Point3, Vec3


class Hittable:
    pass


class Sphere(Hittable):  # sphere.rs:25-45
    def __init__(self, static_center: Point3, radius: float, mat: Material):
        self.center1 = static_center
        self.center2 = static_center
        self.moving = False
        self.radius = float(radius)
        self.mat = mat

    @staticmethod
    def new_moving(center1: Point3, center2: Point3, radius: float, mat: Material) -> "Sphere":
        s = Sphere(center1, radius, mat)
        s.center2 = center2
        s.moving = True
        return s


class Quad(Hittable):  # quad.rs:28-49
    def __init__(self, q: Point3, u: Vec3, v: Vec3, mat: Material):
        self.q, self.u, self.v, self.mat = q, u, v, mat


class Triangle(Hittable):  # triangle.rs:122-149
    def __init__(self, v0: Point3, v1: Point3, v2: Point3, material: Material):
        self.v0, self.v1, self.v2, self.material = v0, v1, v2, material


class HittableList(Hittable):  # hittable_list.rs:15-41
    def __init__(self):
        self.objects: List[Hittable] = []

    @staticmethod
    def with_object(obj: Hittable) -> "HittableList":
        l = HittableList()
        l.add(obj)
        return l

    def clear(self) -> None:
        self.objects.clear()

    def add(self, obj: Hittable) -> None:
        self.objects.append(obj)


class BvhNode(Hittable):  # BvhNode::new(&list) (bvh.rs:14-18); the split happens in the flattener
    def __init__(self, lst: HittableList):
        self.objects = list(lst.objects)


class Translate(Hittable):  # hittable.rs:67-76
    def __init__(self, obj: Hittable, offset: Vec3):
        self.object, self.offset = obj, offset


class RotateY(Hittable):  # hittable.rs:104-143 (angle in degrees)
    def __init__(self, obj: Hittable, angle: float):
        self.object, self.angle = obj, float(angle)


class ConstantMedium(Hittable):  # constant_medium.rs:20-46
    def __init__(self, boundary: Hittable, density: float, phase_function: Material):
        self.boundary, self.density, self.phase_function = boundary, float(density), phase_function

    @staticmethod
    def new_with_texture(boundary: Hittable, density: float, tex: Texture) -> "ConstantMedium":
        return ConstantMedium(boundary, density, Isotropic(tex))

    @staticmethod
    def new_with_color(boundary: Hittable, density: float, albedo: Color) -> "ConstantMedium":
        return ConstantMedium(boundary, density, Isotropic.from_color(albedo))


class TriangleSet(Hittable):
    """Triangles held as one (N, 3, 3) float64 array sharing a material: the bulk form of
    what mesh.rs:126-157 / obj_loader.rs:46-75 build one `Triangle` at a time."""

    def __init__(self, vertices: np.ndarray, material: Material, bvh: bool):
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        if v.ndim != 3 or v.shape[1:] != (3, 3):
            raise ValueError("expected an (N, 3, 3) array of triangle vertices")
        self.vertices, self.material, self.bvh = v, material, bvh


class Mesh(TriangleSet):  # mesh.rs:95-183: BVH over the triangles of an OBJ file
    def __init__(self, vertices: np.ndarray, material: Material):
        super().__init__(vertices, material, bvh=True)

    @staticmethod
    def from_obj(path: str, material: Material) -> "Mesh":  # mesh.rs:101-167
        from .obj_loader import load_obj
        return Mesh(load_obj(path), material)


class ObjModel(TriangleSet):  # obj_loader.rs:13-106: scaled + offset triangles, linear scan
    def __init__(self, vertices: np.ndarray, material: Material):
        super().__init__(vertices, material, bvh=False)

    @staticmethod
    def load(path: str, material: Material, scale: float, offset: Point3) -> "ObjModel":  # obj_loader.rs:21-83
        from .obj_loader import load_obj
        return ObjModel(load_obj(path, scale, (offset.x(), offset.y(), offset.z())), material)  # p * scale + offset per component (obj_loader.rs:51-67)


class SphereSet(Hittable):
    """Static spheres held as arrays (centers (N, 3), radii (N,)) sharing a material; `bvh`
    selects BvhNode::new over them (bvh.rs:14-18) or a plain HittableList."""

    def __init__(self, centers: np.ndarray, radii: np.ndarray, material: Material, bvh: bool = True):
        c = np.ascontiguousarray(centers, dtype=np.float64)
        r = np.ascontiguousarray(radii, dtype=np.float64)
        if c.ndim != 2 or c.shape[1] != 3 or r.shape != (c.shape[0],):
            raise ValueError("expected centers (N, 3) and radii (N,)")
        self.centers, self.radii, self.material, self.bvh = c, r, material, bvh


def box_new(a: Point3, b: Point3, mat: Material) -> HittableList:  # quad.rs:139-193
    sides = HittableList()
    mn = Point3(min(a.x(), b.x()), min(a.y(), b.y()), min(a.z(), b.z()))
    mx = Point3(max(a.x(), b.x()), max(a.y(), b.y()), max(a.z(), b.z()))
    dx = Vec3(mx.x() - mn.x(), 0.0, 0.0)
    dy = Vec3(0.0, mx.y() - mn.y(), 0.0)
    dz = Vec3(0.0, 0.0, mx.z() - mn.z())
    sides.add(Quad(Point3(mn.x(), mn.y(), mx.z()), dx, dy, mat))   # front
    sides.add(Quad(Point3(mx.x(), mn.y(), mx.z()), -dz, dy, mat))  # right
    sides.add(Quad(Point3(mx.x(), mn.y(), mn.z()), -dx, dy, mat))  # back
    sides.add(Quad(Point3(mn.x(), mn.y(), mn.z()), dz, dy, mat))   # left
    sides.add(Quad(Point3(mn.x(), mx.y(), mx.z()), dx, -dz, mat))  # top
    sides.add(Quad(Point3(mn.x(), mn.y(), mn.z()), dx, dz, mat))   # bottom
    return sides
