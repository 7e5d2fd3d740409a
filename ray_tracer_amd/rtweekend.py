"""Host-side utilities of rtweekend.rs.  `rand::thread_rng()` (rtweekend.rs:15-21) is
replaced by the library's counter-based stream 1 (scene construction), so a scene
script builds the same scene on every run and on every rank."""
from __future__ import annotations

from ._ffi import lib

INFINITY = float("inf")  # rtweekend.rs:6
PI = 3.1415926535897932385  # rtweekend.rs:7

_state = {"seed": 0, "index": 0}


def seed(value: int) -> None:
    """Restart the scene-construction stream."""
    _state["seed"] = int(value) & 0xFFFFFFFFFFFFFFFF
    _state["index"] = 0


def degrees_to_radians(degrees: float) -> float:  # rtweekend.rs:11-13
    return degrees * PI / 180.0


def random_double() -> float:  # rtweekend.rs:15-17
    v = lib.rt_rng_uniform(_state["seed"], 1, 0, 0, _state["index"])
    _state["index"] += 1
    return v


def random_double_range(lo: float, hi: float) -> float:  # rtweekend.rs:19-21
    return lo + (hi - lo) * random_double()


def random_int(lo: int, hi: int) -> int:  # rtweekend.rs:23-25 (`as i32` truncates toward zero)
    return int(random_double_range(float(lo), float(hi + 1)))
