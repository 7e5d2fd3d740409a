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


def _float_to_byte(values: np.ndarray) -> np.ndarray:  # rtw_stb_image.rs:134-142
    """f32 channel values -> bytes the way the loader stores them: <= 0 -> 0, >= 1 -> 255, otherwise the f32
    product `value * 255.0` truncated (not rounded).  For a byte b widened as b / 255 the product truncates back to b
    for all 256 values (tests/test_textures_host.py), so the truncation only shows on 16-bit and float sources."""
    v = np.asarray(values, dtype=np.float32)
    scaled = (v * np.float32(255.0)).astype(np.float32)
    out = np.where(v <= 0, 0, np.where(v >= 1, 255, np.trunc(np.nan_to_num(scaled, nan=0.0)))).astype(np.uint8)
    return out


def byte_round_trip_table() -> np.ndarray:
    """What an 8-bit channel value becomes on its way through rtw_stb_image.rs:104-131: the decoder's RGB8 is widened
    to f32 as b / 255 (`to_rgb32f`) and narrowed again by `_float_to_byte`."""
    b = np.arange(256, dtype=np.float32)
    return _float_to_byte(b / np.float32(255.0))


class RtwImage:  # rtw_stb_image.rs:9-16: RGB8 rows, top row first
    SEARCH_DIRS = (".", "images", "../images", "../../images", "../../../images", "../../../../images",
                   "../../../../../images", "../../../../../../images")  # rtw_stb_image.rs:49-57

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

    @classmethod
    def load(cls, image_filename: str) -> "RtwImage":  # rtw_stb_image.rs:23-38
        """Looks for the file under $RTW_IMAGES, then as given, then in images/ up to six levels up; a file that is
        not found or does not decode gives the empty image (height 0: the texture then shades cyan, texture.rs:92)
        after a line on stderr, as the reference does.  Decoding itself is Pillow's, not the `image` crate's: for
        PNG, PPM and baseline JPEG at 8 bits both give the file's samples, but the crate is not in this tree to
        check against."""
        import os
        import sys

        candidates = []
        if os.environ.get("RTW_IMAGES"):
            candidates.append(os.path.join(os.environ["RTW_IMAGES"], image_filename))
        candidates.append(image_filename)
        candidates += [os.path.join(d, image_filename) for d in cls.SEARCH_DIRS[1:]]
        for path in candidates:
            px = cls._decode(path)
            if px is not None:
                return cls(px)
        print(f"ERROR: Could not load image file '{image_filename}'", file=sys.stderr)
        return cls(None)

    @staticmethod
    def _decode(path: str):  # rtw_stb_image.rs:69-116: decode, widen to f32 RGB, narrow to bytes
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise RuntimeError("RtwImage.load needs Pillow to decode image files") from e
        try:
            with Image.open(path) as im:
                im.load()
                if im.mode in ("I;16", "I;16L", "I;16B", "I"):  # 16-bit grey: value / 65535 per channel
                    grey = np.asarray(im, dtype=np.float32) / np.float32(65535.0)
                    f = np.repeat(grey[:, :, None], 3, axis=2)
                elif im.mode == "F":
                    f = np.repeat(np.asarray(im, dtype=np.float32)[:, :, None], 3, axis=2)
                else:                                              # 8-bit: alpha dropped, grey replicated
                    f = np.asarray(im.convert("RGB"), dtype=np.float32) / np.float32(255.0)
        except (OSError, ValueError, SyntaxError):
            return None
        return _float_to_byte(f)


class ImageTexture(Texture):  # texture.rs:75-108
    def __init__(self, image: RtwImage | np.ndarray | str | None):
        if isinstance(image, str):  # texture.rs:81-85: constructed from a file name
            image = RtwImage.load(image)
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
