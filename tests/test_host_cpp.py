"""C++ host mirror (ray_tracer_amd/host): the scene scripts written in C++ against the C ABI must
describe the same scenes as the Python mirror (CPU check) and render the same picture (GPU check)."""
import json
import os
import subprocess

import numpy as np
import pytest

import ray_tracer_amd as rta
from ray_tracer_amd import scenes
from ray_tracer_amd.scene import Scene, SceneDesc

HOST = os.path.join(os.path.dirname(rta.__file__), "host")
CLI = os.path.join(HOST, "rt_cli")

PY_SCENES = {
    "book1_final": lambda seed: scenes.book1_final(seed=seed),
    "quads": lambda seed: scenes.quads(),
    "simple_light": lambda seed: scenes.simple_light(seed=seed),
    "cornell_box": lambda seed: scenes.cornell_box(),
    "cornell_smoke": lambda seed: scenes.cornell_smoke(),
}


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    return CLI


@pytest.mark.parametrize("name", sorted(PY_SCENES))
def test_cpp_scene_description_matches_python(cli, name):
    out = subprocess.run([cli, "--scene", name, "--seed", "5", "--describe"], check=True, capture_output=True, text=True).stdout
    got = json.loads(out)
    world, lights, _ = PY_SCENES[name](5)
    desc = SceneDesc.from_world(world, lights)
    sc = Scene(desc, host_only=True)
    info = sc.info()
    st = desc.struct
    want = {"scene": name, "objects": st.n_objects, "spheres": st.n_spheres, "quads": st.n_quads, "triangles": st.n_triangles, "nodes": info.n_nodes, "lists": info.n_lists, "xforms": info.n_xforms,
            "media": info.n_media, "lights": info.n_lights, "max_stack": info.max_stack}
    assert got == want


def test_cpp_cli_rejects_unknown_scene(cli):
    r = subprocess.run([cli, "--scene", "nope"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,spp,depth", [("cornell_box", 48, 16, 8), ("book1_final", 64, 4, 6), ("cornell_smoke", 40, 9, 6)])
def test_cpp_render_equals_python_render(cli, tmp_path, name, width, spp, depth):
    """Same scene script, same seed, two host languages: the PPM files must be byte-identical."""
    out_cpp = str(tmp_path / "cpp.ppm")
    out_py = str(tmp_path / "py.ppm")
    subprocess.run([cli, "--scene", name, "--seed", "5", "--width", str(width), "--spp", str(spp), "--depth", str(depth),
                    "--out", out_cpp], check=True, capture_output=True, timeout=120)
    world, lights, cam = PY_SCENES[name](5)
    cam.image_width, cam.samples_per_pixel, cam.max_depth = width, spp, depth
    cam.render(world, lights, seed=5, out=out_py)
    a, b = open(out_cpp, "rb").read(), open(out_py, "rb").read()
    assert a[:2] == b"P3" and len(a) > 1000
    assert a == b
