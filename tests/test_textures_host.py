"""Image-file textures on the host side (rtw_stb_image.rs, texture.rs:75-108): search path, the f32 round trip of
the samples, and the empty image a missing file gives.  The decoder here is Pillow, the reference's is the `image`
crate, which is not in the tree: that both give the file's own samples for 8-bit PNG/PPM is assumed, not pinned."""
import os

import numpy as np
import pytest

from ray_tracer_amd.texture import ImageTexture, RtwImage, _float_to_byte, byte_round_trip_table

PIL = pytest.importorskip("PIL.Image")


def test_eight_bit_samples_survive_the_f32_round_trip():
    # rtw_stb_image.rs:104-142: b -> b / 255 (f32) -> (v * 255.0) as u8
    assert np.array_equal(byte_round_trip_table(), np.arange(256, dtype=np.uint8))


def test_float_to_byte_truncates_and_clamps():  # rtw_stb_image.rs:134-142
    v = np.array([-1.0, 0.0, 0.5, 0.999, 1.0, 7.0, 100.9 / 255.0], dtype=np.float32)
    assert _float_to_byte(v).tolist() == [0, 0, 127, 254, 255, 255, 100]


def _write_png(path, h=5, w=7, mode="RGB"):
    rng = np.random.default_rng(3)
    px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    PIL.fromarray(px, "RGB").convert(mode).save(path)
    return px


def test_load_reads_rows_top_first(tmp_path):
    px = _write_png(tmp_path / "t.png")
    img = RtwImage.load(str(tmp_path / "t.png"))
    assert (img.width, img.height) == (7, 5)
    assert np.array_equal(img.pixels, px)
    ppm = tmp_path / "t.ppm"
    PIL.fromarray(px, "RGB").save(ppm)
    assert np.array_equal(RtwImage.load(str(ppm)).pixels, px)


def test_alpha_is_dropped_and_grey_is_replicated(tmp_path):
    px = _write_png(tmp_path / "a.png", mode="RGBA")
    assert np.array_equal(RtwImage.load(str(tmp_path / "a.png")).pixels, px)
    _write_png(tmp_path / "g.png", mode="L")
    g = RtwImage.load(str(tmp_path / "g.png")).pixels
    assert np.array_equal(g[..., 0], g[..., 1]) and np.array_equal(g[..., 1], g[..., 2])


def test_sixteen_bit_grey_is_truncated_not_rounded(tmp_path):
    v = np.array([[0, 255, 256, 257, 32767, 65534, 65535]], dtype=np.uint16)
    PIL.fromarray(v, "I;16").save(tmp_path / "s.png")
    got = RtwImage.load(str(tmp_path / "s.png")).pixels[0, :, 0].tolist()
    want = [int(np.float32(np.float32(x) / np.float32(65535.0)) * np.float32(255.0)) if x < 65535 else 255 for x in v[0].tolist()]
    assert got == want and got[1] == 0 and got[5] == 254


def test_search_path_order(tmp_path, monkeypatch):  # rtw_stb_image.rs:41-66
    env_dir, cwd = tmp_path / "env", tmp_path / "a" / "b"
    (tmp_path / "a" / "images").mkdir(parents=True)
    env_dir.mkdir()
    cwd.mkdir(parents=True)
    up = _write_png(tmp_path / "a" / "images" / "x.png", h=2, w=2)
    monkeypatch.chdir(cwd)
    monkeypatch.delenv("RTW_IMAGES", raising=False)
    assert np.array_equal(RtwImage.load("x.png").pixels, up)          # ../images/x.png
    first = _write_png(env_dir / "x.png", h=3, w=3)
    monkeypatch.setenv("RTW_IMAGES", str(env_dir))
    assert np.array_equal(RtwImage.load("x.png").pixels, first)       # $RTW_IMAGES wins


def test_missing_file_gives_the_empty_image(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("RTW_IMAGES", raising=False)
    tex = ImageTexture("nothing_here.png")
    assert tex.rtw_image.height == 0 and tex.rtw_image.pixels is None   # shades cyan, texture.rs:92
    assert "Could not load image file 'nothing_here.png'" in capsys.readouterr().err
    (tmp_path / "broken.png").write_bytes(b"not an image")
    assert RtwImage.load("broken.png").height == 0
