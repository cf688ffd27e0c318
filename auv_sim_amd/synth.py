"""Synthetic Catalina-like planning worlds ("Synthetic world").

The reference builds its world from geopy/shapely (path_planning/catalina.py:32-119,
sharkOccupancyGrid.py:376-393), which is input generation and out of scope; benches, parity tests
and golden fixtures use these seeded stand-ins instead.  Pure `random.Random(seed)` so the same
world can be rebuilt anywhere (GPU box included) from (seed, sizes).
"""
import random

import numpy as np


def make_world(seed=0, n_obstacles=64, box=(-300.0, -100.0, -100.0, 100.0), cell=10.0,
               n_bins=10, bin_len=50, n_habitats=10, start=None, obst_radius=None,
               hab_radius=(8.0, 20.0), pmax=0.3, polygon=None):
    """Return a dict of fp64 arrays describing one world.

    obstacles [O,3] (x,y,r); habitats [H,3]; polygon [V,2] (the box as a counter-clockwise rectangle, or `polygon`:
    vertices inside the box, or the name of one of POLYGONS, scaled to the box -- the draws are the rectangle world's);
    bins [T,2] (t0,t1) integer-valued; cells [C,4] (minx,miny,maxx,maxy) row-major from the
    bottom-left; prob [T,C]; start (x,y) = box centre unless given.
    """
    rng = random.Random(seed)
    x0, y0, x1, y1 = box
    if start is None:
        start = (0.5 * (x0 + x1), 0.5 * (y0 + y1))
    if obst_radius is None:
        obst_radius = (1.0, 4.0) if n_obstacles <= 64 else (1.0, 3.0)
    obstacles = []
    while len(obstacles) < n_obstacles:
        ox = rng.uniform(x0 + 5.0, x1 - 5.0)
        oy = rng.uniform(y0 + 5.0, y1 - 5.0)
        r = rng.uniform(*obst_radius)
        if (ox - start[0]) ** 2 + (oy - start[1]) ** 2 <= (r + 5.0) ** 2:
            continue
        obstacles.append((ox, oy, r))
    habitats = [(rng.uniform(x0, x1), rng.uniform(y0, y1), rng.uniform(*hab_radius))
                for _ in range(n_habitats)]
    ncol = int(round((x1 - x0) / cell))
    nrow = int(round((y1 - y0) / cell))
    cells = []
    for r_ in range(nrow):
        for c_ in range(ncol):
            cells.append((x0 + c_ * cell, y0 + r_ * cell, x0 + (c_ + 1) * cell, y0 + (r_ + 1) * cell))
    bins = [(float(i * bin_len), float((i + 1) * bin_len)) for i in range(n_bins)]
    prob = [[rng.uniform(0.0, pmax) for _ in range(len(cells))] for _ in range(n_bins)]
    return {
        "seed": seed,
        "box": np.array(box, dtype=np.float64),
        "start": np.array(start, dtype=np.float64),
        "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
        "habitats": np.array(habitats, dtype=np.float64).reshape(-1, 3),
        "polygon": _polygon(polygon, box),
        "bins": np.array(bins, dtype=np.float64).reshape(-1, 2),
        "cells": np.array(cells, dtype=np.float64).reshape(-1, 4),
        "prob": np.array(prob, dtype=np.float64).reshape(n_bins, -1),
        "grid_shape": (nrow, ncol),
    }


# workspace outlines in unit-box coordinates (u, v) -> (x0 + u (x1 - x0), y0 + v (y1 - y0))
POLYGONS = {
    # the reference's Catalina boundary (path_planning/catalina.py:71-73: five lat / lon vertices, ~550 m x 345 m) in local
    # metres, normalised to its bounding box: a slanted convex pentagon (the workspace RRT.exploring really runs in)
    "catalina": ((0.0, 0.690), (0.197, 1.0), (1.0, 0.420), (0.747, 0.0), (0.239, 0.330)),
    # concave: a wedge cut into the top edge down to just above the centre, and a clipped corner (8 vertices)
    "notch": ((0.0, 0.0), (0.85, 0.0), (1.0, 0.2), (1.0, 1.0), (0.62, 1.0), (0.5, 0.56), (0.38, 1.0), (0.0, 1.0)),
}


def _polygon(polygon, box):
    x0, y0, x1, y1 = box
    if polygon is None:
        return np.array([(x0, y0), (x1, y0), (x1, y1), (x0, y1)], dtype=np.float64)
    if isinstance(polygon, str):
        return np.array([(x0 + u * (x1 - x0), y0 + v * (y1 - y0)) for u, v in POLYGONS[polygon]], dtype=np.float64)
    return np.array(polygon, dtype=np.float64).reshape(-1, 2)


def make_rect_world(seed=0, n_obstacles=256, size=200.0, start=(20.0, 20.0), goal=(170.0, 180.0),
                    obst_radius=(1.0, 3.0), origin=(0.0, 0.0)):
    """Planner_RRT (gym_rrt) world: rectangle [0,size]^2, circular obstacles clear of start/goal
    (SURVEY.md section 8(d) config 4).  `origin` translates the whole world (rectangle, start, goal, obstacles; the
    draws are those of the (0, 0) world): the reference's bucket grid ignores the boundary's origin
    (gym_rrt/envs/rrt_dubins.py:115-116), so a translated world exercises its negative-index wrap, its
    'out of the habitat environment bound' return and its IndexError."""
    rng = random.Random(seed)
    obstacles = []
    while len(obstacles) < n_obstacles:
        ox = rng.uniform(5.0, size - 5.0)
        oy = rng.uniform(5.0, size - 5.0)
        r = rng.uniform(*obst_radius)
        if (ox - start[0]) ** 2 + (oy - start[1]) ** 2 <= (r + 5.0) ** 2:
            continue
        if (ox - goal[0]) ** 2 + (oy - goal[1]) ** 2 <= (r + 5.0) ** 2:
            continue
        obstacles.append((ox, oy, r))
    gx, gy = float(origin[0]), float(origin[1])
    obst = np.array(obstacles, dtype=np.float64).reshape(-1, 3)
    obst[:, 0] += gx
    obst[:, 1] += gy
    return {
        "seed": seed,
        "rect": np.array([gx, gy, gx + size, gy + size], dtype=np.float64),
        "start": np.array([start[0] + gx, start[1] + gy], dtype=np.float64),
        "goal": np.array([goal[0] + gx, goal[1] + gy], dtype=np.float64),
        "obstacles": obst,
    }


def make_lattice_world(seed=0, n_obstacles=10, lo=0, hi=490, r_range=(10, 30)):
    """astar.astar world (config 1): integer obstacle centres in [50, hi-50], integer radii."""
    rng = random.Random(seed)
    obstacles = []
    while len(obstacles) < n_obstacles:
        ox = rng.randint(50, hi - 50)
        oy = rng.randint(50, hi - 50)
        r = rng.randint(*r_range)
        # keep the start and goal corners free
        if (ox - lo) ** 2 + (oy - lo) ** 2 <= (r + 15) ** 2:
            continue
        if (ox - hi) ** 2 + (oy - hi) ** 2 <= (r + 15) ** 2:
            continue
        obstacles.append((ox, oy, r))
    return {
        "seed": seed,
        "box": np.array([lo, lo, hi, hi], dtype=np.float64),
        "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
        "start": (lo, lo),
        "goal": (hi, hi),
    }
