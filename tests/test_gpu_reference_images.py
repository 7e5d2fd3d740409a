"""The only outputs the reference itself holds: rendered PPM images.  tests/golden/ref_ppm_stats.json carries the
mean colour and an 8 x 8 grid of block means of three of them (made by tests/golden/make_ref_ppm_stats.py from
/root/reference/output28.ppm, output33.ppm, output35.ppm).  Here the same scenes are rendered on the GPU with the
estimator those revisions used (no light list: material PDF alone) and compared statistically.

What this pins: the scene scripts of ray_tracer_amd/scenes.py, the camera, the materials and the look of the
estimator, as evidence NOT authored by this build.  What it cannot pin: bits -- the reference drew from
thread_rng and the Book-2 final scene places 1400 random boxes and spheres, so only colour statistics over
blocks of thousands of pixels are compared, with tolerances in 8-bit units stated per scene."""
import json
import os

import numpy as np
import pytest

import ray_tracer_amd as rt
from ray_tracer_amd import scenes

pytestmark = pytest.mark.gpu

STATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_ppm_stats.json")))


def _blocks(img, grid=8):
    h, w, _ = img.shape
    return np.array([[img[h * j // grid:h * (j + 1) // grid, w * i // grid:w * (i + 1) // grid].mean(axis=(0, 1))
                      for i in range(grid)] for j in range(grid)])


def _render(build):
    world, lights, cam = build
    accum = rt.Scene(rt.SceneDesc.from_world(world, lights)).render(cam.to_struct(), seed=12)
    return rt.Image(cam, accum).rgb.astype(np.float64)


# (scene, tolerance of the mean colour per channel, tolerance of a block mean, share of the 64 blocks within it)
CASES = [
    # Five flat quads under the sky.  output28.ppm predates the camera's `background` field (it appears with the next
    # scene, time_record.txt:14-15): that revision shaded misses with the Book-1 gradient white -> (0.5, 0.7, 1.0),
    # whose blue is the constant 1.0 of today's background (0.7, 0.8, 1.0) while red and green average 0.05 higher.
    # Blue therefore has to agree closely, block by block; red and green carry that offset (up to 40 units near the
    # bottom of the frame, where the gradient is white) and are held to the frame mean only.
    ("quads", (9.0, 8.0, 1.0), 1.5, 1.0),              # measured: blue mean within 0.03, every blue block within 0.3
    ("cornell_two_blocks", (2.0, 2.0, 2.0), 8.0, 1.0),    # noisy at the reference's 200 spp (small light, no light sampling); measured 0.9 / 5.8
    ("cornell_smoke", (2.0, 2.0, 2.0), 7.0, 1.0),         # measured 0.7 / 4.6
    ("book2_final_simple", (2.0, 2.0, 2.0), 25.0, 0.85),  # random box heights / sphere cloud differ from the reference's draw; mean measured 0.6
]


@pytest.mark.parametrize("name,mean_tol,block_tol,block_frac", CASES)
def test_render_statistics_match_the_reference_images(name, mean_tol, block_tol, block_frac):
    ref = STATS[name]
    if name == "quads":
        world, lights, cam = scenes.quads()
        cam.image_width, cam.samples_per_pixel, cam.max_depth = 400, 100, 50
        img = _render((world, lights, cam))
    elif name == "cornell_two_blocks":
        img = _render(scenes.cornell_two_blocks(image_width=600, samples_per_pixel=200, max_depth=50))
    elif name == "cornell_smoke":
        world, lights, cam = scenes.cornell_smoke(image_width=600, samples_per_pixel=200, max_depth=50)
        img = _render((world, None, cam))  # the revision that rendered output34.ppm had no light list yet
    else:
        img = _render(scenes.book2_final(image_width=400, samples_per_pixel=250, max_depth=4))
    assert img.shape == (ref["height"], ref["width"], 3)
    mean = img.mean(axis=(0, 1))
    want = np.array(ref["mean_rgb"])
    print(name, "mean", mean.round(2), "reference", want)
    assert np.all(np.abs(mean - want) <= np.array(mean_tol)), (name, mean.round(2), want)
    diff = np.abs(_blocks(img) - np.array(ref["block_means"]))
    diff = diff[:, :, 2] if name == "quads" else diff.max(axis=2)
    print(name, "largest block differences", np.sort(diff.reshape(-1))[-8:].round(1))
    assert (diff <= block_tol).mean() >= block_frac, (name, np.sort(diff.reshape(-1))[-8:].round(1))
