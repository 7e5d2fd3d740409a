"""Replays chosen scenes of a fuzz_scenes.py campaign and prints the rays on which library and oracle differ
(not collected by pytest: run by name on a GPU).
   python tests/fuzz_repro.py <seed> <rich|plain> <scene> [<scene> ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import _ffi as F, scenes
import oracle_ffi as O
import fuzz_scenes as Z


def main():
    seed = int(sys.argv[1]); rich = sys.argv[2] == "rich"
    wanted = sorted(int(a) for a in sys.argv[3:])
    g = np.random.default_rng(seed)
    m = rt.Lambertian(rt.Color(.5, .5, .5))
    Z.RICH["on"] = rich
    image = rt.ImageTexture(scenes.procedural_image(32, 16))
    for k in range(wanted[-1] + 1):
        Z.RICH["phase"] = image if (rich and k % 2 == 0) else None
        world = rt.HittableList()
        for _ in range(int(g.integers(1, 8))):
            world.add(Z.rand_node(g, m, 0, True))
        try:
            desc = rt.SceneDesc.from_world(world)
            rt.Scene(desc, host_only=True)
        except Exception:
            continue
        ray_seed = int(g.integers(1 << 30))
        if k not in wanted:
            continue
        o, d, tm = scenes.incoherent_rays(6000, [-5, 5, -5, 5, -5, 5], seed=ray_seed)
        gpu = rt.Scene(desc); cpu = O.OracleScene(desc)
        print("scene", k, "max_stack", gpu.info().max_stack, flush=True)
        Z.describe(world, 0)
        for flags in (0, F.RT_HIT_ORDERED):
            a = gpu.closest_hit(o, d, time=tm, seed=0, flags=flags)
            b = cpu.closest_hit(o, d, time=tm, seed=0)
            bad = np.nonzero(~np.isclose(a["u"], b["u"], rtol=0, atol=1e-12, equal_nan=True) | ~np.isclose(a["v"], b["v"], rtol=0, atol=1e-12, equal_nan=True))[0]
            print(" flags", flags, "rays that differ in u,v:", bad.size)
            for i in bad[:8]:
                print("  ray", int(i), "prim gpu", hex(int(a["prim"][i])), "oracle", hex(int(b["prim"][i])), "t", a["t"][i], b["t"][i],
                      "u", a["u"][i], b["u"][i], "v", a["v"][i], b["v"][i])
            if bad.size:
                i = int(bad[0])
                oo = np.repeat(o[i:i + 1], 64, axis=0); dd = np.repeat(d[i:i + 1], 64, axis=0); tt = np.repeat(tm[i:i + 1], 64)
                a1 = gpu.closest_hit(oo, dd, time=tt, seed=0, flags=flags)
                b1 = cpu.closest_hit(oo, dd, time=tt, seed=0)
                print("  the same ray 64 times (ray index = random stream): u gpu", np.unique(a1["u"]), "oracle", np.unique(b1["u"]),
                      "differ", int((~np.isclose(a1["u"], b1["u"], rtol=0, atol=1e-12)).sum()))
        gpu.close()


if __name__ == "__main__":
    main()
