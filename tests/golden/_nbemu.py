"""numba-typing emulator for golden minting (build container only).

TEST INFRASTRUCTURE.  The reference's fit kernels are ``@numba.jit`` functions; numba types
their scalar arithmetic differently from NumPy 2 (NEP 50) AND, in three places, from the legacy
NumPy 1.x scalar promotion as well.  This module executes the reference's own source files from
where they lie under /root/reference with every *jitted* function's arithmetic re-typed by numba's
rules, so that the minted vectors are what the production (numba) path computes, up to the
third-party pieces listed at the bottom.  Nothing of the reference's text is stored: the files are
parsed in memory, the jitted function bodies get their ``a <op> b`` nodes rewritten into calls of
``binop`` below, and the tree is compiled and executed.

Rules (numba 0.6x, read from its published typing tables; each is exercised by a golden):

  scalar (op) scalar    Python int -> int64, Python float -> float64, then the common type:
                        float32 (op) float32 -> float32, float32 (op) int64 -> float64,
                        float32 (op) float64 -> float64, int64 / int64 -> float64
                        (numba/core/typing/builtins.py, BinOp templates: the cheapest SAFE
                        conversion wins).  NumPy applies exactly this to two *NumPy* scalars, in
                        1.x and in 2.x, so after lifting the Python scalars the native operator
                        is used.
  float ** int          square-and-multiply in the float's own type, float32 ** int -> float32
                        ("Ensure that float32 ** int doesn't go through DP computations",
                        builtins.py BinOpPower; numba/cpython/numbers.py int_power_impl /
                        static_power_impl); a negative exponent is 1 / (a ** -b).
  array (op) scalar     NO value-based casting: the ufunc loop is matched on the dtypes
                        (numba/np/numpy_support.py ufunc_find_matching_loop); with mixed integer
                        and float inputs any integer may be cast to the float type
                        (ufunc_can_cast), so float32[:] - int64 -> float32[:], while
                        float64 * float32[:] -> float64[:] (legacy NumPy: float32[:]).
  np.exp / np.log       on a scalar or inside an array expression: the C library's exp / log /
                        expf / logf (llvm.exp.* lowers to libm), not NumPy's SIMD kernels; routed
                        to libm through ctypes.  math.erf is the C library's in CPython and numba.
  array ** 2 (float64)  the 'dd->d' power loop = libm pow(x, 2.0) per element.

NOT emulated (third party, adjudicated in DESIGN.md section 2): numba's ``np.linalg.pinv`` (LAPACK
gesdd + its own product) — NumPy's pinv runs instead; numba raising ZeroDivisionError on a float
division by zero (error_model="python") — IEEE semantics run instead and the event is recorded in
``ZERO_DIVISIONS`` so the goldens can flag the rows the production path would not finish.
"""
from __future__ import annotations

import ast
import ctypes
import ctypes.util
import importlib.util
import math
import operator
import os
import sys
import types
import warnings

import numpy as np

REF = os.environ.get("PICASSO_REFERENCE", "/root/reference")

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("exp", "log", "pow"):
    getattr(_libm, _n).restype = ctypes.c_double
    getattr(_libm, _n).argtypes = [ctypes.c_double] * (2 if _n == "pow" else 1)
for _n in ("expf", "logf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]

ZERO_DIVISIONS = [0]          # float divisions by zero seen since the caller last reset it

_OPS = {
    "Add": operator.add, "Sub": operator.sub, "Mult": operator.mul, "Div": operator.truediv,
    "FloorDiv": operator.floordiv, "Mod": operator.mod, "Pow": operator.pow,
    "BitAnd": operator.and_, "BitOr": operator.or_, "BitXor": operator.xor,
    "LShift": operator.lshift, "RShift": operator.rshift, "MatMult": operator.matmul,
}
_FLOATS = (np.float32, np.float64)


def _lift(v):
    if type(v) is int:
        return np.int64(v)
    if type(v) is float:
        return np.float64(v)
    return v


def _int_power(a, b):
    """numba/cpython/numbers.py int_power_impl: a ** b for an integer b, in a's own type."""
    tp = type(a)
    b = int(b)
    invert = b < 0
    e = -b if invert else b
    r = tp(1)
    while e:
        if e & 1:
            r = tp(r * a)
        e >>= 1
        a = tp(a * a)
    return tp(tp(1) / r) if invert else r


def _array_dtype(a, b):
    da = a.dtype if isinstance(a, np.ndarray) else np.dtype(type(a))
    db = b.dtype if isinstance(b, np.ndarray) else np.dtype(type(b))
    if da.kind in "iub" and db.kind == "f":
        return db
    if db.kind in "iub" and da.kind == "f":
        return da
    return np.promote_types(da, db)


def binop(name, a, b):
    op = _OPS[name]
    a, b = _lift(a), _lift(b)
    arr_a, arr_b = isinstance(a, np.ndarray) and a.ndim > 0, isinstance(b, np.ndarray) and b.ndim > 0
    num_a = arr_a or isinstance(a, np.generic)
    num_b = arr_b or isinstance(b, np.generic)
    if not (num_a and num_b):
        return op(a, b)                                   # strings, tuples, lists ...
    if arr_a or arr_b:
        dt = _array_dtype(a, b)
        if name == "Div" and dt.kind in "iub":
            dt = np.dtype(np.float64)
        x, y = np.asarray(a, dtype=dt), np.asarray(b, dtype=dt)
        if name == "Pow":
            if dt == np.float64:
                out = np.empty(np.broadcast(x, y).shape, np.float64)
                xb, yb = np.broadcast_to(x, out.shape), np.broadcast_to(y, out.shape)
                flat = out.reshape(-1)
                for i, (p, q) in enumerate(zip(xb.reshape(-1), yb.reshape(-1))):
                    flat[i] = _libm.pow(float(p), float(q))
                return out
            return np.power(x, y)
        return op(x, y)
    # scalar (op) scalar
    if name == "Pow" and isinstance(b, np.integer):
        if isinstance(a, _FLOATS) or isinstance(a, np.integer):
            return _int_power(a, b)
    if name == "Div" and isinstance(a, _FLOATS + (np.integer,)) and b == 0:
        ZERO_DIVISIONS[0] += 1
    with np.errstate(all="ignore"):
        return op(a, b)


def _libm_unary(f64, f32):
    def fn(x, *args, **kwargs):
        if args or kwargs:
            raise TypeError("emulated np.exp / np.log take one argument")
        x = _lift(x)
        if isinstance(x, np.ndarray) and x.ndim > 0:
            if x.dtype == np.float32:
                return np.array([f32(float(v)) for v in x.reshape(-1)], np.float32).reshape(x.shape)
            return np.array([f64(float(v)) for v in x.reshape(-1).astype(np.float64)], np.float64).reshape(x.shape)
        if isinstance(x, np.float32):
            return np.float32(f32(float(x)))
        return np.float64(f64(float(x)))
    return fn


class _NumpyProxy:
    """``np`` as the jitted functions see it: libm for exp / log, NumPy for the rest."""
    exp = staticmethod(_libm_unary(_libm.exp, _libm.expf))
    log = staticmethod(_libm_unary(_libm.log, _libm.logf))

    def __getattr__(self, name):
        return getattr(np, name)


class _Retype(ast.NodeTransformer):
    """Inside @numba.jit / njit / vectorize functions: a <op> b -> __nb_binop__(op, a, b)."""

    def __init__(self):
        self.depth = 0
        self.rewritten = []

    @staticmethod
    def _is_jitted(node):
        for d in node.decorator_list:
            f = d.func if isinstance(d, ast.Call) else d
            if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == "numba" \
                    and f.attr in ("jit", "njit", "vectorize"):
                return True
        return False

    def visit_FunctionDef(self, node):
        jitted = self._is_jitted(node)
        if jitted:
            self.depth += 1
            self.rewritten.append(node.name)
        node.body = [self.visit(s) for s in node.body]
        if jitted:
            self.depth -= 1
        return node

    def _call(self, op, left, right, ref):
        call = ast.Call(func=ast.Name(id="__nb_binop__", ctx=ast.Load()),
                        args=[ast.Constant(value=type(op).__name__), left, right], keywords=[])
        return ast.copy_location(call, ref)

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if not self.depth:
            return node
        return self._call(node.op, node.left, node.right, node)

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if not self.depth:
            return node
        load = ast.parse(ast.unparse(node.target), mode="eval").body      # the target as a Load expression
        ast.copy_location(load, node)
        for n in ast.walk(load):
            ast.copy_location(n, node)
        # a nested subscript of the target may itself hold arithmetic: retype it too
        load = self.visit(load) if not isinstance(load, ast.Name) else load
        value = self._call(node.op, load, node.value, node)
        return ast.copy_location(ast.Assign(targets=[node.target], value=value), node)


def _stub_modules():
    def identity(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    numba = types.ModuleType("numba")
    numba.jit = numba.njit = identity
    numba.vectorize = lambda *a, **k: (np.vectorize(a[0]) if len(a) == 1 and callable(a[0]) and not k
                                       else (lambda fn: np.vectorize(fn)))
    numba.prange = range
    sys.modules["numba"] = numba

    pkg = types.ModuleType("picasso")
    pkg.__path__ = [os.path.join(REF, "picasso")]
    pkg.__version__ = "0.10.3"
    sys.modules["picasso"] = pkg
    lib = types.ModuleType("picasso.lib")
    lib.__getattr__ = lambda name: object                 # annotation-only names
    lib.deprecation_warning = lambda message: None
    sys.modules["picasso.lib"] = lib
    pkg.lib = lib
    ext = types.ModuleType("picasso.ext")
    ext.__path__ = []                                     # `from .ext.pygpufit import gpufit` fails -> feature flag False
    sys.modules["picasso.ext"] = ext
    for name in ("tqdm",):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.tqdm = lambda it, **k: it
                sys.modules[name] = m


def load(name):
    """Execute picasso/<name>.py from the reference tree with numba's typing inside its jitted functions."""
    if not os.path.isdir(os.path.join(REF, "picasso")):
        raise RuntimeError(f"reference tree not found at {REF}")
    _stub_modules()
    path = os.path.join(REF, "picasso", name + ".py")
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    tr = _Retype()
    tree = ast.fix_missing_locations(tr.visit(tree))
    code = compile(tree, path, "exec")
    mod = types.ModuleType("picasso." + name)
    mod.__file__ = path
    mod.__package__ = "picasso"
    mod.__dict__["__nb_binop__"] = binop
    sys.modules["picasso." + name] = mod
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(code, mod.__dict__)
    mod.__dict__["np"] = _NumpyProxy()                    # seen by the functions at call time
    mod.__retyped__ = tr.rewritten
    return mod
