"""Host mirror of texture.rs / perlin.rs / rtw_stb_image.rs: plain descriptions that the
scene builder turns into rt_texture / rt_perlin / rt_image records.  Evaluation happens in
the shade kernel."""
from __future__ import annotations

import numpy as np

from . import rtweekend
from .vec3 import Color, Vec3, unit_vector


class Texture:
    pass


class SolidColor(Texture):  # texture.rs:11-31
    def __init__(self, albedo: Color):
        self.albedo = albedo

    @staticmethod
    def from_rgb(red: float, green: float, blue: float) -> "SolidColor":
        return SolidColor(Color(red, green, blue))


class CheckerTexture(Texture):  # texture.rs:34-72
    def __init__(self, scale: float, even: Texture, odd: Texture):
        self.scale = float(scale)
        self.even = even
        self.odd = odd

    @staticmethod
    def from_colors(scale: float, c1: Color, c2: Color) -> "CheckerTexture":
        return CheckerTexture(scale, SolidColor(c1), SolidColor(c2))


class RtwImage:  # rtw_stb_image.rs: RGB8 rows; decoding image files is outside the hot path
    def __init__(self, pixels: np.ndarray | None):
        if pixels is None:
            self.pixels = None
            self.width = self.height = 0
        else:
            px = np.ascontiguousarray(pixels, dtype=np.uint8)
            if px.ndim != 3 or px.shape[2] != 3:
                raise ValueError("RtwImage expects an (H, W, 3) uint8 array")
            self.pixels = px
            self.height, self.width = int(px.shape[0]), int(px.shape[1])


class ImageTexture(Texture):  # texture.rs:75-108
    def __init__(self, image: RtwImage | np.ndarray | None):
        self.rtw_image = image if isinstance(image, RtwImage) else RtwImage(image)


class Perlin:  # perlin.rs:7-53 (tables only; noise()/turb() run on the device)
    POINT_COUNT = 256

    def __init__(self):
        self.randvec = []
        for _ in range(self.POINT_COUNT):
            self.randvec.append(unit_vector(Vec3.random_range(-1.0, 1.0)))
        self.perm_x = self._generate_perm()
        self.perm_y = self._generate_perm()
        self.perm_z = self._generate_perm()

    @classmethod
    def _generate_perm(cls):  # perlin.rs:39-53
        p = list(range(cls.POINT_COUNT))
        for i in range(cls.POINT_COUNT - 1, 0, -1):
            target = rtweekend.random_int(0, i)
            p[i], p[target] = p[target], p[i]
        return p


class NoiseTexture(Texture):  # texture.rs:111-130
    def __init__(self, scale: float):
        self.noise = Perlin()
        self.scale = float(scale)
