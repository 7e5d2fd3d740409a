import time, sys
sys.path.insert(0, "/root/repo")
import torch
import ray_tracer_amd as rt
from ray_tracer_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
kind = sys.argv[2] if len(sys.argv) > 2 else "soup"
w = scenes.triangle_soup_scene(n, seed=2) if kind == "soup" else rt.Mesh(scenes.icosphere_mesh(subdiv=8), rt.Lambertian(rt.Color(.5, .5, .5)))  # closed mesh: 1,310,720 triangles
desc = rt.SceneDesc.from_world(w)
gpu = torch.cuda.is_available()
for i in range(3):
    t0 = time.perf_counter()
    s = rt.Scene(desc, host_only=not gpu)
    print("build", time.perf_counter() - t0, flush=True)
    s.close()
