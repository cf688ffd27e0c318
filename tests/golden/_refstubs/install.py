"""Oracle-harness stubs: lets the *reference* planners be imported in the build container.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py (and nothing else) to import
/root/reference in this container, where shapely / geopy / descartes / gym are not installed
(SURVEY.md section 8(c)).  Never shipped, never imported by the product package, never used on
the GPU box (the reference does not exist there).

The stubs restate only the third-party behaviour the hot path touches:
  * shapely.geometry.Polygon   .bounds / .exterior.xy / .exterior.coords / .centroid.coords
  * shapely.geometry.Point     .within(polygon)  -> even-odd crossing test, strict comparisons
    (this stub *is* the definition of "inside" for the build; the reference pins nothing here,
    see DESIGN.md "parity unpinned at the shapely boundary")
  * geopy.distance.vincenty    value only feeds a discarded accumulator in astar_fixLen*.py
  * gym                        import-only
"""
import sys
import types


class _Coords(list):
    @property
    def xy(self):
        return [c[0] for c in self], [c[1] for c in self]

    @property
    def coords(self):
        return list(self)


class Polygon:
    def __init__(self, coords):
        pts = [(float(p[0]), float(p[1])) for p in coords]
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts = pts[:-1]
        self._pts = pts

    @property
    def bounds(self):
        xs = [p[0] for p in self._pts]
        ys = [p[1] for p in self._pts]
        return (min(xs), min(ys), max(xs), max(ys))

    @property
    def exterior(self):
        return _Coords(self._pts + [self._pts[0]])

    @property
    def centroid(self):
        # area-weighted centroid (shoelace), the published definition of Polygon.centroid
        pts = self._pts
        n = len(pts)
        a2 = 0.0
        cx = 0.0
        cy = 0.0
        for i in range(n):
            x0, y0 = pts[i]
            x1, y1 = pts[(i + 1) % n]
            cr = x0 * y1 - x1 * y0
            a2 += cr
            cx += (x0 + x1) * cr
            cy += (y0 + y1) * cr
        c = types.SimpleNamespace()
        c.coords = [(cx / (3.0 * a2), cy / (3.0 * a2))]
        c.x = c.coords[0][0]
        c.y = c.coords[0][1]
        return c


class Point:
    def __init__(self, x, y):
        self.x = float(x)
        self.y = float(y)

    def within(self, poly):
        # even-odd crossing number; j trails i by one vertex
        pts = poly._pts
        n = len(pts)
        inside = False
        j = n - 1
        for i in range(n):
            xi, yi = pts[i]
            xj, yj = pts[j]
            if (yi > self.y) != (yj > self.y):
                if self.x < (xj - xi) * (self.y - yi) / (yj - yi) + xi:
                    inside = not inside
            j = i
        return inside


class CellStub:
    """stand-in for a shapely cell polygon: only .bounds is read by the planners"""

    def __init__(self, minx, miny, maxx, maxy):
        self.bounds = (minx, miny, maxx, maxy)
        self._pts = [(minx, miny), (maxx, miny), (maxx, maxy), (minx, maxy)]

    @property
    def exterior(self):
        return _Coords(self._pts + [self._pts[0]])

    def touches(self, pt):
        """shapely: the point lies on the cell's boundary (sharkOccupancyGrid.py:264 uses
        `point.within(cell) or cell.touches(point)` = closed containment)"""
        minx, miny, maxx, maxy = self.bounds
        on_x = (pt.x == minx or pt.x == maxx) and miny <= pt.y <= maxy
        on_y = (pt.y == miny or pt.y == maxy) and minx <= pt.x <= maxx
        return on_x or on_y


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Register the stub modules in sys.modules (idempotent)."""
    if "shapely" in sys.modules and getattr(sys.modules["shapely"], "_auvp_stub", False):
        return

    def _na(*a, **k):
        raise NotImplementedError("not needed on the hot path")

    geom = _mod("shapely.geometry", Polygon=Polygon, Point=Point, box=_na, MultiPolygon=_na,
                GeometryCollection=_na, LinearRing=_na, LineString=_na)
    ops = _mod("shapely.ops", split=_na)
    wkt = _mod("shapely.wkt", loads=_na)
    _mod("shapely", geometry=geom, ops=ops, wkt=wkt, _auvp_stub=True)

    _mod("descartes", PolygonPatch=_na)

    class _V:
        m = 0.0

        def __init__(self, *a, **k):
            pass

    dist = _mod("geopy.distance", vincenty=_V, distance=_V)
    _mod("geopy", distance=dist)

    # gym: import-only
    class _Env:
        pass

    class _Space:
        def __init__(self, *a, **k):
            self.args = a
            self.kwargs = k

    spaces = _mod("gym.spaces", Dict=_Space, Box=_Space, Discrete=_Space, MultiDiscrete=_Space,
                  Tuple=_Space)
    seeding = _mod("gym.utils.seeding", np_random=lambda seed=None: (None, seed))
    utils = _mod("gym.utils", seeding=seeding)
    error = _mod("gym.error")
    reg = _mod("gym.envs.registration", register=lambda *a, **k: None)
    envs = _mod("gym.envs", registration=reg)
    _mod("gym", Env=_Env, spaces=spaces, utils=utils, error=error, envs=envs)

    import matplotlib
    matplotlib.use("Agg")
    matplotlib.use = lambda *a, **k: None  # gym_rrt/envs/rrt_dubins.py:6 forces TkAgg


class VirtualClock:
    """Deterministic stand-in for the `time` module inside path_planning/rrt_dubins.py.

    time() returns max(0, c-2) where c counts calls made from a frame named `exploring`
    (the t_start read at rrt_dubins.py:107 gives 0; the k-th `while` test at :118 gives k-1);
    calls from any other frame (steer :252,:284) return the current value without advancing.
    With plot_interval == max_plan_time == n_iter this gives exactly n_iter loop iterations and
    plan_time_stamp == 0-based iteration index.
    """

    def __init__(self):
        self.c = 0

    def time(self):
        f = sys._getframe(1)
        if f.f_code.co_name in ("exploring", "planning"):
            self.c += 1
        return float(max(0, self.c - 2))

    def sleep(self, s):
        pass
