"""Stand-in modules that let tests/golden/make_golden.py EXECUTE the reference's own Python statements in this container,
where casadi, shapely, gymnasium and matplotlib are not installed (ordinary ModuleNotFoundError, no network).  Used only by
the fixture generators (which need /root/reference); nothing here is imported by the product, by bench.py or by a test.

What each one is, and what fixtures made through it therefore do and do not pin:

`casadi`  - a NUMERIC evaluator, not a symbolic library and not a solver.  `SX.sym(name, r, c)` returns a matrix holding the
    float64 values the generator put into `POINT[name]`; sin / cos / tan / atan / vertcat / reshape / norm_2 / if_else and
    the operators evaluate with numpy.  The reference's statements that build its objective, constraints, bounds and initial
    guess (agents/pure_mpc.py:128-283, agents/archive/pure_mpc.py:118-288) therefore compute f(z), g(z), lbx, ubx, x0 at the
    point z the generator chose; `Function(...)` records the cost components it is given and `nlpsol(...)` hands back an
    object whose call records its arguments and raises `Captured` - there is no IPOPT here, nothing is solved.
    Pins: the NLP definition at the points evaluated.  Does not pin: the solver (a11).
`shapely` - `LineString(coords).intersection(other)` for an arbitrary polyline against a polyline whose vertices are collinear
    (the constant-velocity prediction of agents/pure_mpc.py:529-550), built from the segment arithmetic of
    tests/host_preamble.py and returned in the shapes the reference branches on (agents/pure_mpc.py:615-633): empty, Point,
    LineString (a collinear overlap: ONE line with all noded coordinates, along the first geometry), MultiPoint /
    MultiLineString (members along the first geometry), GeometryCollection for a mix.  GEOS's own noding, merging and
    member order cannot be reproduced - that stays a documented deviation.
    Pins: everything the reference does AROUND that call - candidate loop, sample indices, conflict index, the 10-step
    memory, the speed-profile rewrite.  Does not pin: GEOS.
`gymnasium`, `matplotlib` - empty (type annotations / plotting only).
"""
import sys
import types

import numpy as np

POINT = {}          # name -> ndarray (r, c): the values SX.sym hands out
CAPTURED = {}       # filled by Function / nlpsol / the solver call


class Captured(Exception):
    """Raised by the solver stand-in once the NLP and the call arguments are recorded."""


def _val(o):
    if isinstance(o, NumSX):
        return o.a
    a = np.asarray(o, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)        # casadi reads a 1-D numpy array as a column
    return a


class NumSX:
    """Dense float64 matrix with casadi's SX surface as far as the reference uses it."""
    __array_ufunc__ = None          # numpy scalars defer to the reflected operators below
    __array_priority__ = 1000

    def __init__(self, a):
        a = np.array(a, dtype=np.float64)
        self.a = a.reshape(1, 1) if a.ndim == 0 else (a.reshape(-1, 1) if a.ndim == 1 else a)

    @staticmethod
    def sym(name, r, c=1):
        v = np.asarray(POINT[name], dtype=np.float64)
        assert v.shape == (r, c), (name, v.shape, (r, c))
        return NumSX(v.copy())

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx, 0) if self.a.shape[1] == 1 else (idx,)
        rows = idx[0]
        cols = idx[1] if len(idx) > 1 else slice(None)
        sub = self.a[rows if isinstance(rows, slice) else slice(rows, rows + 1 if rows != -1 else None),
                     cols if isinstance(cols, slice) else slice(cols, cols + 1 if cols != -1 else None)]
        return NumSX(sub)

    def size1(self):
        return self.a.shape[0]

    def __float__(self):
        assert self.a.size == 1
        return float(self.a[0, 0])

    def __bool__(self):
        assert self.a.size == 1
        return bool(self.a[0, 0])

    def full(self):
        return self.a.copy()

    def __neg__(self):
        return NumSX(-self.a)

    def __add__(self, o):
        return NumSX(self.a + _val(o))

    __radd__ = __add__

    def __sub__(self, o):
        return NumSX(self.a - _val(o))

    def __rsub__(self, o):
        return NumSX(_val(o) - self.a)

    def __mul__(self, o):
        return NumSX(self.a * _val(o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return NumSX(self.a / _val(o))

    def __rtruediv__(self, o):
        return NumSX(_val(o) / self.a)

    def __pow__(self, p):
        return NumSX(self.a * self.a) if p == 2 else NumSX(self.a ** p)

    def __lt__(self, o):
        return NumSX((self.a < _val(o)).astype(np.float64))


def _fn(f):
    def g(x):
        return NumSX(f(x.a)) if isinstance(x, NumSX) else float(f(x))
    return g


def _vertcat(*args):
    return NumSX(np.vstack([_val(a) if not np.isscalar(a) else np.array([[float(a)]]) for a in args]))


def _reshape(x, r, c):
    return NumSX(_val(x).reshape((r, c), order="F"))       # casadi is column-major


def _if_else(cond, a, b):
    c = _val(cond) if isinstance(cond, NumSX) else np.array([[1.0 if cond else 0.0]])
    return NumSX(np.where(c != 0.0, _val(a), _val(b)))


def _norm_2(x):
    v = _val(x)
    return NumSX(np.sqrt(np.sum(v * v)))


class _Function:
    def __init__(self, name, args, outs, *a, **k):
        CAPTURED.setdefault("functions", {})[name] = [float(_val(o).ravel()[0]) if not np.isscalar(o) else float(o) for o in outs]


class _Solver:
    def __init__(self, name, plugin, nlp, opts=None):
        CAPTURED["plugin"] = plugin
        CAPTURED["opts"] = dict(opts or {})
        CAPTURED["f"] = float(_val(nlp["f"]).ravel()[0])
        CAPTURED["g"] = _val(nlp["g"]).ravel().copy()
        CAPTURED["x"] = _val(nlp["x"]).ravel().copy()

    def __call__(self, **kw):
        for k, v in kw.items():
            CAPTURED[k] = np.asarray([float(t) for t in np.ravel(v)], dtype=np.float64)
        if SOLVE_HOOK[0] is None:
            raise Captured()
        # a solution handed in by the fixture generator (z in the reference's layout [X.ravel(), U.ravel()], success flag):
        # the reference's statements BEHIND the nlpsol call (agents/pure_mpc.py:300-318) then run on it
        z, ok = SOLVE_HOOK[0](CAPTURED)
        self._ok = bool(ok)
        return {"x": _DM(np.asarray(z, np.float64).reshape(-1, 1))}

    def stats(self):
        return {"success": self._ok}


class _DM:
    """What casadi hands back as sol['x']: a column that slices like one and converts with .full()"""

    def __init__(self, a):
        self.a = np.asarray(a, np.float64).reshape(-1, 1)

    def full(self):
        return self.a.copy()

    def __getitem__(self, k):
        return _DM(self.a[k])


SOLVE_HOOK = [None]


# ---------------------------------------------------------------------------------------------------------------
class _Geom:
    is_empty = False


class _Empty(_Geom):
    is_empty = True
    geom_type = "GeometryCollection"


class _Point(_Geom):
    geom_type = "Point"

    def __init__(self, p):
        self.x, self.y = float(p[0]), float(p[1])


class _Line(_Geom):
    geom_type = "LineString"

    def __init__(self, coords):
        self.coords = [(float(c[0]), float(c[1])) for c in coords]


class _Multi(_Geom):
    def __init__(self, geom_type, geoms):
        self.geom_type, self.geoms = geom_type, list(geoms)


class LineString:
    def __init__(self, coords):
        self.pts = np.asarray([np.asarray(c, dtype=np.float64) for c in coords])
        if len(self.pts) == 1:      # shapely 2.0: "point array must contain 0 or >1 elements" (what :582-587 catches)
            raise sys.modules["shapely.errors"].GEOSException("IllegalArgumentException: point array must contain 0 or >1 elements")

    def intersection(self, other):
        from host_preamble import path_pieces          # tests/ is on the generator's sys.path
        pieces = path_pieces(self.pts, other.pts)
        if not pieces:
            return _Empty()
        geoms = [_Point(p) if kind == "point" else _Line(p) for kind, p in pieces]
        if len(geoms) == 1:
            return geoms[0]
        kinds = {g.geom_type for g in geoms}
        if kinds == {"Point"}:
            return _Multi("MultiPoint", geoms)
        if kinds == {"LineString"}:
            return _Multi("MultiLineString", geoms)
        return _Multi("GeometryCollection", geoms)


# ---------------------------------------------------------------------------------------------------------------
# cvxpy: the same idea as the casadi stand-in - a NUMERIC evaluator.  Variable((r, c)) holds the values the generator put
# into POINT["cvx"] (in creation order); quad_form / abs / the operators evaluate with numpy; `==`, `<=`, `>=` return records
# of the residual (equalities) or slack (inequalities, >= 0 when satisfied); Problem(...).solve() stores the objective value
# and those records in CAPTURED and reports a status that makes the reference take its "solver failed" exit
# (agents/pure_mpc_linear.py:260-262) - nothing is solved.  Pins: the QP of _linear_mpc_control (:205-257) at the points
# evaluated.  Does not pin: ECOS.
class CvxExpr:
    __array_ufunc__ = None
    __array_priority__ = 1000

    def __init__(self, a):
        self.a = np.asarray(a, dtype=np.float64)

    @staticmethod
    def v(o):
        return o.a if isinstance(o, CvxExpr) else np.asarray(o, dtype=np.float64)

    def __getitem__(self, idx):
        return CvxExpr(self.a[idx])

    def __add__(self, o):
        return CvxExpr(self.a + CvxExpr.v(o))

    __radd__ = __add__

    def __sub__(self, o):
        return CvxExpr(self.a - CvxExpr.v(o))

    def __rsub__(self, o):
        return CvxExpr(CvxExpr.v(o) - self.a)

    def __mul__(self, o):
        return CvxExpr(self.a * CvxExpr.v(o))

    __rmul__ = __mul__

    def __neg__(self):
        return CvxExpr(-self.a)

    def __matmul__(self, o):
        return CvxExpr(self.a @ CvxExpr.v(o))

    def __rmatmul__(self, o):
        return CvxExpr(CvxExpr.v(o) @ self.a)

    def __eq__(self, o):
        return ("eq", np.ravel(self.a - CvxExpr.v(o)).copy())

    def __le__(self, o):
        return ("le", np.ravel(CvxExpr.v(o) - self.a).copy())

    def __ge__(self, o):
        return ("le", np.ravel(self.a - CvxExpr.v(o)).copy())

    __hash__ = None


class _CvxVariable(CvxExpr):
    def __init__(self, shape):
        vals = POINT["cvx"].pop(0)
        assert vals.shape == tuple(shape), (vals.shape, shape)
        super().__init__(vals.copy())
        self.value = None


class _CvxProblem:
    def __init__(self, objective, constraints):
        self.objective, self.constraints, self.status = objective, constraints, "not_solved"

    def solve(self, **kw):
        CAPTURED["cvx_cost"] = float(np.ravel(CvxExpr.v(self.objective))[0])
        CAPTURED["cvx_eq"] = np.concatenate([r for k, r in self.constraints if k == "eq"])
        CAPTURED["cvx_le"] = np.concatenate([r for k, r in self.constraints if k == "le"])
        CAPTURED["cvx_solver"] = kw.get("solver")
        self.status = "stand-in: nothing solved"
        return None


def install():
    """Put the stand-ins into sys.modules (only names that are really absent are replaced)."""
    for name in ("gymnasium", "casadi", "shapely", "shapely.errors", "matplotlib", "matplotlib.pyplot", "cvxpy"):
        try:
            __import__(name)
            continue
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["gymnasium"].Env = object
    sh = sys.modules["shapely"]
    sh.LineString = LineString
    sh.errors = sys.modules["shapely.errors"]
    sys.modules["shapely.errors"].GEOSException = type("GEOSException", (Exception,), {})
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    ca = sys.modules["casadi"]
    ca.SX = NumSX
    ca.sin, ca.cos, ca.tan, ca.atan = _fn(np.sin), _fn(np.cos), _fn(np.tan), _fn(np.arctan)
    ca.vertcat, ca.reshape, ca.if_else, ca.norm_2 = _vertcat, _reshape, _if_else, _norm_2
    ca.Function, ca.nlpsol = _Function, _Solver
    ca.pi, ca.inf = float(np.pi), float("inf")
    cp = sys.modules["cvxpy"]
    cp.Variable = _CvxVariable
    cp.quad_form = lambda x, Q: CvxExpr(CvxExpr.v(x) @ np.asarray(Q, np.float64) @ CvxExpr.v(x))
    cp.abs = lambda x: CvxExpr(np.abs(CvxExpr.v(x)))
    cp.Minimize = lambda c: c
    cp.Problem = _CvxProblem
    cp.ECOS, cp.OPTIMAL, cp.OPTIMAL_INACCURATE = "ECOS", "optimal", "optimal_inaccurate"
