"""Host mirror of material.rs: descriptions consumed by the scene builder; scatter,
emitted and scattering_pdf run in the shade kernel."""
from __future__ import annotations

from .texture import SolidColor, Texture
from .vec3 import Color


class Material:
    pass


class Lambertian(Material):  # material.rs:58-75
    def __init__(self, albedo: Color):
        self.tex: Texture = SolidColor(albedo)

    @staticmethod
    def from_texture(texture: Texture) -> "Lambertian":
        m = Lambertian.__new__(Lambertian)
        m.tex = texture
        return m


class Metal(Material):  # material.rs:120-133
    def __init__(self, albedo: Color, fuzz: float):
        self.albedo = albedo
        self.fuzz = min(float(fuzz), 1.0)


class Dielectric(Material):  # material.rs:161-172
    def __init__(self, refraction_index: float):
        self.refraction_index = float(refraction_index)


class DiffuseLight(Material):  # material.rs:220-235
    def __init__(self, tex: Texture):
        self.tex = tex

    @staticmethod
    def from_color(emit: Color) -> "DiffuseLight":
        return DiffuseLight(SolidColor(emit))


class Isotropic(Material):  # material.rs:261-276
    def __init__(self, tex: Texture):
        self.tex = tex

    @staticmethod
    def from_color(albedo: Color) -> "Isotropic":
        return Isotropic(SolidColor(albedo))
