"""Host-side Vec3 / Point3 / Color (the reference's src/vec3.rs is missing from the
snapshot; this is the API its call sites use: camera.rs:8, material.rs:10-12, pdf.rs:3,7,
quad.rs:11, onb.rs:1, main.rs:134,145,757).  f64 throughout, `v / t` is `v * (1/t)`."""
from __future__ import annotations

import math

from . import rtweekend


class Vec3:
    __slots__ = ("e",)

    def __init__(self, x: float = 0.0, y: float = 0.0, z: float = 0.0):
        self.e = (float(x), float(y), float(z))

    def x(self) -> float:
        return self.e[0]

    def y(self) -> float:
        return self.e[1]

    def z(self) -> float:
        return self.e[2]

    def __getitem__(self, i: int) -> float:
        return self.e[i]

    def __iter__(self):
        return iter(self.e)

    def __neg__(self) -> "Vec3":
        return Vec3(-self.e[0], -self.e[1], -self.e[2])

    def __add__(self, o: "Vec3") -> "Vec3":
        return Vec3(self.e[0] + o.e[0], self.e[1] + o.e[1], self.e[2] + o.e[2])

    def __sub__(self, o: "Vec3") -> "Vec3":
        return Vec3(self.e[0] - o.e[0], self.e[1] - o.e[1], self.e[2] - o.e[2])

    def __mul__(self, o) -> "Vec3":
        if isinstance(o, Vec3):
            return Vec3(self.e[0] * o.e[0], self.e[1] * o.e[1], self.e[2] * o.e[2])
        t = float(o)
        return Vec3(t * self.e[0], t * self.e[1], t * self.e[2])

    __rmul__ = __mul__

    def __truediv__(self, t: float) -> "Vec3":
        return self * (1.0 / float(t))

    def length_squared(self) -> float:
        return self.e[0] * self.e[0] + self.e[1] * self.e[1] + self.e[2] * self.e[2]

    def length(self) -> float:
        return math.sqrt(self.length_squared())

    def min(self, o: "Vec3") -> "Vec3":
        return Vec3(min(self.e[0], o.e[0]), min(self.e[1], o.e[1]), min(self.e[2], o.e[2]))

    def max(self, o: "Vec3") -> "Vec3":
        return Vec3(max(self.e[0], o.e[0]), max(self.e[1], o.e[1]), max(self.e[2], o.e[2]))

    @staticmethod
    def random() -> "Vec3":
        x = rtweekend.random_double()
        y = rtweekend.random_double()
        z = rtweekend.random_double()
        return Vec3(x, y, z)

    @staticmethod
    def random_range(lo: float, hi: float) -> "Vec3":
        x = rtweekend.random_double_range(lo, hi)
        y = rtweekend.random_double_range(lo, hi)
        z = rtweekend.random_double_range(lo, hi)
        return Vec3(x, y, z)

    def __repr__(self) -> str:
        return f"Vec3({self.e[0]!r}, {self.e[1]!r}, {self.e[2]!r})"


Point3 = Vec3
Color = Vec3  # color.rs:5


def dot(a: Vec3, b: Vec3) -> float:
    return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]


def cross(a: Vec3, b: Vec3) -> Vec3:
    return Vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0])


def unit_vector(v: Vec3) -> Vec3:
    return v / v.length()
