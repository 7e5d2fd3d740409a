"""Multi-process path on the CPU: gloo groups of 2, 3 and 5 ranks run the frame partition + sum-reduce of
ray_tracer_amd.parallel with the ORACLE as the per-rank renderer, and the result must equal the
unsplit frame.  (The GPU renderer plugs into the same functions; its partition independence is
covered by tests/test_gpu_render.py.)"""
import os
import socket

import numpy as np
import pytest

from ray_tracer_amd import _ffi as F
from ray_tracer_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_covers_frame_exactly_once():
    for mode in ("strata", "tiles", "auto"):
        for world in (1, 2, 3, 4, 8):
            for (w, h, ns) in ((40, 30, 16), (7, 5, 4), (600, 600, 961), (16, 3, 1)):
                cover = np.zeros((h, w, ns), dtype=np.int32)
                for rank in range(world):
                    for u in parallel.partition_frame(w, h, ns, rank, world, mode):
                        assert 0 <= u.x0 < u.x1 <= w and 0 <= u.y0 < u.y1 <= h and 0 <= u.sample_begin < u.sample_end <= ns
                        cover[u.y0:u.y1, u.x0:u.x1, u.sample_begin:u.sample_end] += 1
                assert np.all(cover == 1), (mode, world, w, h, ns)
    for world in (2, 8):
        pieces = [parallel.shard_rays(1000003, r, world) for r in range(world)]
        assert pieces[0][0] == 0 and pieces[-1][1] == 1000003
        assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in pieces) - min(b - a for a, b in pieces) <= 1


def _worker(rank, world, port, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import oracle_ffi as O
    import ray_tracer_amd as rt
    from ray_tracer_amd import scenes

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        world_obj, lights, cam = scenes.cornell_box(image_width=24, samples_per_pixel=9, max_depth=6)
        desc = rt.SceneDesc.from_world(world_obj, lights)
        cpu = O.OracleScene(desc)
        c = cam.to_struct()
        d = O.camera_initialize(c)
        accum = torch.zeros((d.image_height, cam.image_width, 3), dtype=torch.float64)

        def render_unit(work, acc):
            cpu.render(c, seed=11, work=work, accum=acc.numpy(), nthreads=2)

        parallel.render_distributed(render_unit, accum, cam.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, mode=mode, dist=dist)
        # the timed job of bench.py --gpus N (same partition, same reduce), on a bounded sample of the strata
        acc2 = torch.zeros_like(accum)
        acc2, tm = parallel.render_frame_job(render_unit, acc2, cam.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, dist=dist,
                                             mode=mode, strata_limit=5)
        assert tm["world_size"] == world and tm["rank"] == rank and tm["strata"] == 5 and tm["render_s"] >= 0 and tm["reduce_s"] >= 0
        # the other exchange of each partition: reduce-scatter by band owner + gather for strata, a plain sum for tiles
        other = "reduce_scatter" if mode == "strata" else "reduce"
        acc3 = torch.zeros_like(accum)
        acc3, tm3 = parallel.render_frame_job(render_unit, acc3, cam.image_width, d.image_height, d.sqrt_spp * d.sqrt_spp, dist=dist,
                                              mode=mode, strata_limit=5, collective=other)
        assert tm["collective"] == ("reduce" if mode == "strata" else "gather") and tm3["collective"] == other
        counts = [None] * world
        dist.all_gather_object(counts, tm["paths"])
        if rank == 0:
            assert sum(counts) == cam.image_width * d.image_height * 5  # every (pixel, stratum) of the sample exactly once
            full = cpu.render(c, seed=11, nthreads=2)
            part = cpu.render(c, seed=11, work=F.rt_work(0, 0, cam.image_width, d.image_height, 0, 5), nthreads=2)
            np.save(out_path, np.stack([accum.numpy(), full, acc2.numpy(), part, acc3.numpy()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("strata", 2), ("tiles", 2), ("strata", 5), ("tiles", 3)])
def test_multi_rank_render_equals_unsplit_frame(tmp_path, mode, world):
    """World sizes 2, 3 and 5: with five ranks the accumulator's 1728 numbers do not divide into equal bands (the
    reduce-scatter pads), and the 24 image rows of the tiles partition split unevenly over three."""
    import torch.multiprocessing as mp
    out = str(tmp_path / f"r_{mode}_{world}.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    got, full, got5, part5, other5 = np.load(out)
    assert full.mean() > 1e-3
    assert np.allclose(np.nan_to_num(got), np.nan_to_num(full), rtol=1e-12, atol=1e-14)
    assert np.allclose(np.nan_to_num(got5), np.nan_to_num(part5), rtol=1e-12, atol=1e-14)
    assert np.allclose(np.nan_to_num(other5), np.nan_to_num(part5), rtol=1e-12, atol=1e-14)


def test_a_gather_cannot_serve_the_strata_partition():
    with pytest.raises(ValueError):
        parallel.render_frame_job(lambda w, a: None, None, 8, 8, 4, dist=None, mode="strata", collective="gather")
