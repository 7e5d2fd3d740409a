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
    if dist is not None:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def _band(numel: int, world: int) -> int:
    """Elements per rank when a flat accumulator of `numel` elements is cut into `world` equal bands (the last one padded)."""
    return (numel + world - 1) // world


def collect_accumulator(accum, dist, collective: str = "reduce", dst: int = 0, mode: str = "strata"):
    """The frame-end exchange.  `accum`: this rank's (H, W, 3) float64 accumulator, on its device.

    "reduce":          one sum-reduction of the whole accumulator to rank `dst` (a ring over xGMI: every link carries
                       about the accumulator's size once).
    "reduce_scatter":  reduce-scatter by band owner, then a gather of the bands to `dst`: every rank ends up with the
                       sum of ITS band of image rows (each pair of ranks exchanges one band directly, all links busy),
                       and `dst` collects world-1 bands.  Backends without reduce_scatter (gloo) run it as one reduce
                       per band owner: the same result.
    "gather":          for the "tiles" partition, where every pixel was rendered by exactly one rank: no sum, the
                       bands of rows go to `dst` as they are.
    Returns the accumulator (meaningful on `dst`).  With one rank the collective still runs (a no-op in effect), so
    that the call path is the one a multi-rank job takes.
    """
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if collective == "reduce":
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
        return accum
    flat = accum.reshape(-1)
    band = _band(flat.numel(), world)
    if collective == "gather":
        # rows [y0, y1) of rank r are contiguous in the flat accumulator; bands are padded to one size for the gather
        h = accum.shape[0]
        row = flat.numel() // max(1, h)
        rows = [_split(h, world, r) for r in range(world)]
        width = max(y1 - y0 for y0, y1 in rows) * row
        y0, y1 = rows[rank]
        mine = torch.zeros(width, dtype=accum.dtype, device=accum.device)
        mine[:(y1 - y0) * row] = flat[y0 * row:y1 * row]
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, parts, dst=dst)
        if rank == dst:
            for r, (a, b) in enumerate(rows):
                flat[a * row:b * row] = parts[r][:(b - a) * row]
        return accum
    if collective != "reduce_scatter":
        raise ValueError(f"unknown collective {collective!r}")
    padded = flat
    if band * world != flat.numel():
        padded = torch.zeros(band * world, dtype=accum.dtype, device=accum.device)
        padded[:flat.numel()] = flat
    mine = torch.empty(band, dtype=accum.dtype, device=accum.device)
    if dist.get_backend() == "nccl":
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM)
    else:
        for owner in range(world):
            piece = padded[owner * band:(owner + 1) * band].clone()
            dist.reduce(piece, dst=owner, op=dist.ReduceOp.SUM)
            if owner == rank:
                mine = piece
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst)
    if rank == dst:
        whole = torch.cat(parts)[:flat.numel()]
        flat.copy_(whole)
    return accum


def render_frame_job(render_unit: Callable[[F.rt_work, "object"], None], accum, width: int, height: int, n_strata: int,
                     dist=None, mode: str = "strata", strata_limit: Optional[int] = None, sync: Optional[Callable[[], None]] = None,
                     dst: int = 0, collective: Optional[str] = None):
    """One frame as a multi-rank job, timed: this rank's units of the partition, then the frame-end exchange of the
    accumulator (collect_accumulator; default: "reduce" for the strata partition, "gather" for tiles).
    `strata_limit` renders only strata [0, limit) of the camera's grid (a bounded sample of a frame whose full
    sample count would take minutes); `sync` blocks until the rank's device is idle (torch.cuda.synchronize for
    the GPU renderer, nothing for a CPU renderer).  Returns (accum, timings) where timings = {"rank", "world_size",
    "strata", "paths", "render_s", "reduce_s", "collective"}; the same function serves the benchmark (RCCL) and the
    CPU tests (gloo).  The exchange runs whenever a process group is given, with one rank too.
    """
    import time
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    used = n_strata if strata_limit is None else max(0, min(n_strata, int(strata_limit)))
    if collective is None:
        collective = "gather" if mode == "tiles" else "reduce"
    if collective == "gather" and mode != "tiles":
        raise ValueError("a gather only serves the tiles partition: strata of one pixel are spread over the ranks and must be summed")
    sync = sync or (lambda: None)
    sync()
    t0 = time.perf_counter()
    paths = 0
    for work in partition_frame(width, height, used, rank, world, mode):
        render_unit(work, accum)
        paths += (work.x1 - work.x0) * (work.y1 - work.y0) * (work.sample_end - work.sample_begin)
    sync()
    t1 = time.perf_counter()
    if dist is not None:
        collect_accumulator(accum, dist, collective, dst=dst, mode=mode)
        sync()
    t2 = time.perf_counter()
    return accum, {"rank": rank, "world_size": world, "strata": used, "paths": paths, "render_s": t1 - t0, "reduce_s": t2 - t1,
                   "collective": collective if dist is not None else None}


def shard_rays(n: int, rank: int, world_size: int) -> tuple:
    """Contiguous ray range of `rank` for a closest-hit batch of n rays (no exchange; hits stay local
    or are gathered by the caller)."""
    return _split(n, world_size, rank)
