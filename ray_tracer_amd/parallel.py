"""Frame partition across GPUs and the frame-end reduction of the radiance accumulators.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on
CPU for tests).  The scene is replicated; work is split into (tile, stratum-range) units
(`rt_work`).  Paths are independent and every random stream is keyed by (pixel, stratum)
(camera.rs:99-131 has no cross-pixel state), so the frame does not depend on the partition;
the only exchange is ONE sum-reduction of the H x W x 3 f64 accumulator to rank 0 at frame end.
"""
from __future__ import annotations

from typing import Callable, List, Optional

from . import _ffi as F


def _split(n: int, parts: int, index: int) -> tuple:
    """Contiguous near-equal split of range(n): the `index`-th of `parts` pieces as (begin, end)."""
    base, extra = divmod(n, parts)
    begin = index * base + min(index, extra)
    return begin, begin + base + (1 if index < extra else 0)


def partition_frame(width: int, height: int, n_strata: int, rank: int, world_size: int, mode: str = "auto") -> List[F.rt_work]:
    """Work units of `rank`.

    "strata": every rank renders the whole image for a contiguous band of the sqrt_spp^2 strata
              (perfect load balance; needs the sum).
    "tiles":  every rank renders all strata of a band of image rows (content-dependent load).
    "auto":   strata when there are at least as many strata as ranks, else tiles.
    """
    if not (0 <= rank < world_size):
        raise ValueError("rank outside the world")
    if mode == "auto":
        mode = "strata" if n_strata >= world_size else "tiles"
    if mode == "strata":
        s0, s1 = _split(n_strata, world_size, rank)
        return [F.rt_work(0, 0, width, height, s0, s1)] if s1 > s0 else []
    if mode == "tiles":
        y0, y1 = _split(height, world_size, rank)
        return [F.rt_work(0, y0, width, y1, 0, n_strata)] if y1 > y0 else []
    raise ValueError(f"unknown partition mode {mode!r}")


def render_distributed(render_unit: Callable[[F.rt_work, "object"], None], accum, width: int, height: int, n_strata: int,
                       mode: str = "auto", dist=None, dst: int = 0):
    """Renders this rank's units into `accum` (a zero-initialised (H, W, 3) float64 torch tensor on the
    rank's device) through `render_unit(work, accum)` and sum-reduces it to rank `dst`.

    `render_unit` is `lambda w, a: scene.render_device(cam, a, seed, w, stream)` in production; the CPU
    tests pass the oracle's renderer.  Returns `accum` (meaningful on `dst`).
    """
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    for work in partition_frame(width, height, n_strata, rank, world, mode):
        render_unit(work, accum)
    if dist is not None and world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def render_frame_job(render_unit: Callable[[F.rt_work, "object"], None], accum, width: int, height: int, n_strata: int,
                     dist=None, mode: str = "strata", strata_limit: Optional[int] = None, sync: Optional[Callable[[], None]] = None,
                     dst: int = 0):
    """One frame as a multi-rank job, timed: this rank's units of the partition, then the one sum-reduction of the
    accumulator to rank `dst`.  `strata_limit` renders only strata [0, limit) of the camera's grid (a bounded
    sample of a frame whose full sample count would take minutes); `sync` blocks until the rank's device is idle
    (torch.cuda.synchronize for the GPU renderer, nothing for a CPU renderer).  Returns (accum, timings) where
    timings = {"rank", "world_size", "strata", "paths", "render_s", "reduce_s"}; the same function serves the
    benchmark (RCCL) and the CPU tests (gloo).
    """
    import time
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    used = n_strata if strata_limit is None else max(0, min(n_strata, int(strata_limit)))
    sync = sync or (lambda: None)
    sync()
    t0 = time.perf_counter()
    paths = 0
    for work in partition_frame(width, height, used, rank, world, mode):
        render_unit(work, accum)
        paths += (work.x1 - work.x0) * (work.y1 - work.y0) * (work.sample_end - work.sample_begin)
    sync()
    t1 = time.perf_counter()
    if dist is not None and world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
        sync()
    t2 = time.perf_counter()
    return accum, {"rank": rank, "world_size": world, "strata": used, "paths": paths, "render_s": t1 - t0, "reduce_s": t2 - t1}


def shard_rays(n: int, rank: int, world_size: int) -> tuple:
    """Contiguous ray range of `rank` for a closest-hit batch of n rays (no exchange; hits stay local
    or are gathered by the caller)."""
    return _split(n, world_size, rank)
