"""A serial, pure-Python stand-in for the `taichi` module -- TEST INFRASTRUCTURE.

Purpose: Taichi cannot be installed in this image, so the reference
(/root/reference/*.py, Python that only runs under Taichi) cannot be executed as
is.  This package implements just enough of the Taichi API for the reference's
OWN, UNMODIFIED source files (particle_system.py, sph_base.py, WCSPH.py, DFSPH.py,
config_builder.py) to be imported and run on the CPU, one loop iteration at a
time, with f32 / i32 arithmetic done by NumPy scalars.  oracle/gen_golden.py uses
it to produce the golden vectors under tests/golden/ that pin oracle/sph_oracle.c.

What it reproduces: every formula, the traversal order of every loop (a Taichi
parallel `for` is executed serially in index order, which is also the order that
makes the reference's atomic-rank counting sort stable), ti.template() pass-by-
reference, atomic_add/sub return values, Vector/Matrix value semantics.
What it cannot reproduce: the Taichi compiler's own rounding choices (fast-math,
pow lowering, constant folding of Python-scope values) and ti.polar_decompose's
approximate SVD (NumPy's exact SVD is used).

It is NOT a general Taichi emulator.
"""
from __future__ import annotations

import ast
import builtins
import inspect
import itertools
import textwrap
import types as _pytypes

import numpy as np

f32 = np.float32
i32 = np.int32
f64 = np.float64
cuda = "cuda"
vulkan = "vulkan"
gpu = "gpu"
cpu = "cpu"

oob_reads = 0  # out-of-range field reads (undefined behaviour in the reference); generators assert it stays 0


def init(*a, **k):
    return None


def data_oriented(cls):
    return cls


def static(x):
    return x


def template():
    return _Template()


class _Template:
    pass


class _Types:
    @staticmethod
    def ndarray(*a, **k):
        return "ndarray"

    @staticmethod
    def vector(*a, **k):
        return "vector"

    @staticmethod
    def matrix(*a, **k):
        return "matrix"


types = _Types()


def _np_dtype(dt):
    if dt in (float, f32, "f32"):
        return np.float32
    if dt in (int, i32, "i32"):
        return np.int32
    if dt is f64:
        return np.float64
    return np.dtype(dt).type


def cast(v, dt):
    return _np_dtype(dt)(v)


# ---------------------------------------------------------------------------
# value types
# ---------------------------------------------------------------------------
class Vec(np.ndarray):
    """ti.Vector / ti.Matrix value (f32 or i32).  A Vec read from a field keeps a
    back-reference so `field[i][k] = v` and `field[i].fill(v)` write through."""
    _owner = None

    def __new__(cls, data, dtype=None):
        arr = np.asarray(data)
        if dtype is None:
            dtype = np.int32 if arr.dtype.kind in "iub" else np.float32
        return np.array(arr, dtype=dtype).view(cls)

    def __array_finalize__(self, obj):
        self._owner = None

    def __array_wrap__(self, out, context=None, return_scalar=False):
        out = np.ndarray.__array_wrap__(self, out, context, return_scalar) if not return_scalar else out
        return out

    def norm(self):
        return np.sqrt(np.sum(self * self, dtype=self.dtype))

    def norm_sqr(self):
        return np.sum(np.asarray(self) * np.asarray(self), dtype=self.dtype)

    def dot(self, other):
        return np.sum(np.asarray(self) * np.asarray(other), dtype=self.dtype)

    def outer_product(self, other):
        return Vec(np.outer(np.asarray(self), np.asarray(other)), dtype=self.dtype)

    def cast(self, dt):
        return Vec(np.trunc(np.asarray(self)) if _np_dtype(dt) is np.int32 else np.asarray(self), dtype=_np_dtype(dt))

    def normalized(self):
        return self / self.norm()

    def fill(self, v):
        np.ndarray.fill(self, v)
        if self._owner is not None:
            fld, idx = self._owner
            fld.arr[idx] = self

    def __setitem__(self, k, v):
        np.ndarray.__setitem__(self, k, v)
        if self._owner is not None:
            fld, idx = self._owner
            fld.arr[idx] = self

    def to_numpy(self):
        return np.array(self)


def _vec_ctor(data, dt=None):
    return Vec(data, None if dt is None else _np_dtype(dt))


class _VectorNS:
    def __call__(self, data, dt=None):
        return _vec_ctor(data, dt)

    @staticmethod
    def zero(dt, n, m=None):
        return Vec(np.zeros(n if m is None else (n, m)), _np_dtype(dt))

    @staticmethod
    def field(n, dtype=float, shape=None):
        return Field(_np_dtype(dtype), shape, (n,))

    @staticmethod
    def identity(dt, n):
        return Vec(np.identity(n), _np_dtype(dt))


Vector = _VectorNS()
Matrix = _VectorNS()


def _idx(i):
    if i is None:
        return ()
    if isinstance(i, np.ndarray):
        return tuple(int(v) for v in i.reshape(-1))
    if isinstance(i, tuple):
        return tuple(int(v) for v in i)
    return (int(i),)


class Field:
    def __init__(self, dtype, shape, elem_shape=()):
        if shape is None:
            shape = ()
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        self.elem_shape = tuple(elem_shape)
        self.dtype = dtype
        self.arr = np.zeros(self.shape + self.elem_shape, dtype=dtype)

    def _check(self, idx):
        global oob_reads
        for k, s in zip(idx, self.shape):
            if k < 0 or k >= s:
                oob_reads += 1
                return False
        return True

    def __getitem__(self, i):
        idx = _idx(i)
        if not self._check(idx):
            return self.dtype(0) if not self.elem_shape else Vec(np.zeros(self.elem_shape), self.dtype)
        v = self.arr[idx]
        if self.elem_shape:
            out = Vec(v, self.dtype)
            out._owner = (self, idx)
            return out
        return self.dtype(v)

    def __setitem__(self, i, v):
        idx = _idx(i)
        if not self._check(idx):
            raise IndexError(f"field store out of range: {idx} for shape {self.shape}")
        self.arr[idx] = v

    def to_numpy(self):
        return self.arr.copy()

    def from_numpy(self, a):
        self.arr[...] = a

    def fill(self, v):
        self.arr[...] = v


def field(dtype=float, shape=None):
    return Field(_np_dtype(dtype), shape)


def grouped(x):
    if isinstance(x, Field):
        return (Vec(ix, np.int32) for ix in itertools.product(*[range(s) for s in x.shape]))
    return (Vec(ix, np.int32) for ix in x)


def ndrange(*args):
    rs = [range(a[0], a[1]) if isinstance(a, tuple) else range(a) for a in args]
    return itertools.product(*rs)


def pow(a, b):  # noqa: A001
    return np.power(a, b, dtype=np.float32) if not isinstance(a, Vec) else a ** b


def max(*a):  # noqa: A001
    return builtins.max(*a)


def min(*a):  # noqa: A001
    return builtins.min(*a)


def sqrt(x):
    return np.sqrt(x)


def abs(x):  # noqa: A001
    return np.abs(_unbox(x))


class Struct:
    """ti.Struct(**members): a mutable record (members keep NumPy f32 / Python-weak scalar semantics)."""

    def __init__(self, **members):
        self.__dict__.update(members)


def polar_decompose(A):
    a = np.asarray(A, dtype=np.float64)
    U, s, Vt = np.linalg.svd(a)
    if np.linalg.det(U) < 0:
        U[:, -1] *= -1
        s[-1] *= -1
    if np.linalg.det(Vt) < 0:
        Vt[-1, :] *= -1
        s[-1] *= -1
    return Vec(U @ Vt, np.float32), Vec(Vt.T @ np.diag(s) @ Vt, np.float32)


class _PrefixSumExecutor:
    def __init__(self, n):
        self.n = n

    def run(self, fld):
        fld.arr[...] = np.cumsum(fld.arr, dtype=np.int32)  # inclusive, in place, i32


class _Algorithms:
    PrefixSumExecutor = _PrefixSumExecutor


algorithms = _Algorithms()


class _NS:
    def __getattr__(self, name):
        return _NS()

    def __call__(self, *a, **k):
        raise RuntimeError("taichi shim: SIMT intrinsics are not emulated")


simt = _NS()
ui = _NS()
tools = _NS()
profiler = _NS()


def global_thread_idx():
    raise RuntimeError("not emulated")


def loop_config(**k):
    return None


# ---------------------------------------------------------------------------
# kernel / func: AST rewrite for pass-by-reference templates and atomics
# ---------------------------------------------------------------------------
class Box:
    """Cell that carries a ti.template() argument by reference.  `x += ...` on the parameter name rebinds the
    cell's value (see _aug); reads of a boxed scalar, `ret[k] += ...` on a boxed Vector and `ret.member += ...` on
    a boxed Struct go through to the value."""
    __slots__ = ("v",)

    def __init__(self, v):
        object.__setattr__(self, "v", v)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "v"), name)

    def __setattr__(self, name, value):
        if name == "v":
            object.__setattr__(self, "v", value)
        else:
            setattr(self.v, name, value)

    def __getitem__(self, k):
        return self.v[k]

    def __setitem__(self, k, value):
        self.v[k] = value

    def __add__(self, o): return self.v + _unbox(o)
    def __radd__(self, o): return _unbox(o) + self.v
    def __sub__(self, o): return self.v - _unbox(o)
    def __rsub__(self, o): return _unbox(o) - self.v
    def __mul__(self, o): return self.v * _unbox(o)
    def __rmul__(self, o): return _unbox(o) * self.v
    def __truediv__(self, o): return self.v / _unbox(o)
    def __rtruediv__(self, o): return _unbox(o) / self.v
    def __neg__(self): return -self.v
    def __abs__(self): return np.abs(self.v)
    def __lt__(self, o): return self.v < _unbox(o)
    def __le__(self, o): return self.v <= _unbox(o)
    def __gt__(self, o): return self.v > _unbox(o)
    def __ge__(self, o): return self.v >= _unbox(o)
    def __float__(self): return float(self.v)


def _unbox(x):
    return x.v if isinstance(x, Box) else x


_OPS = {ast.Add: lambda a, b: a + b, ast.Sub: lambda a, b: a - b, ast.Mult: lambda a, b: a * b,
        ast.Div: lambda a, b: a / b}


def _aug(target, opname, value):
    op = _OPS[getattr(ast, opname)]
    if isinstance(target, Box):
        target.v = op(target.v, value)
        return target
    return op(target, value)


def _atomic(container, index, value, sign):
    old = container[index]
    container[index] = old + value if sign > 0 else old - value
    return old


def atomic_add(x, v):  # only reachable if the AST rewrite missed a pattern
    raise RuntimeError("taichi shim: atomic_add target must be a subscript expression")


atomic_sub = atomic_add


def _callee_template_flags(f):
    w = getattr(f, "_ti_wrapper", None)
    if w is None:
        return None
    return w.template_flags


def _box_args(f, args):
    flags = _callee_template_flags(f)
    if flags is None:
        return list(args)
    out = []
    for k, a in enumerate(args):
        is_t = k < len(flags) and flags[k]
        if is_t and not isinstance(a, (Box, Field)) and not callable(a):
            out.append(Box(a))
        else:
            out.append(a)
    return out


def _after(orig, new):
    if isinstance(orig, Box):
        return orig
    if isinstance(new, Box):
        return new.v
    return orig


class _Rewriter(ast.NodeTransformer):
    def __init__(self):
        self.counter = 0

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.target, ast.Name) and type(node.op) in _OPS:
            call = ast.Call(func=ast.Name("__ti_aug", ast.Load()),
                            args=[ast.Name(node.target.id, ast.Load()), ast.Constant(type(node.op).__name__),
                                  node.value], keywords=[])
            return ast.copy_location(ast.Assign(targets=[ast.Name(node.target.id, ast.Store())], value=call), node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if (isinstance(f, ast.Attribute) and f.attr in ("atomic_add", "atomic_sub") and isinstance(f.value, ast.Name)
                and f.value.id == "ti" and node.args and isinstance(node.args[0], ast.Subscript)):
            sub = node.args[0]
            return ast.copy_location(ast.Call(
                func=ast.Name("__ti_atomic", ast.Load()),
                args=[sub.value, sub.slice, node.args[1], ast.Constant(1 if f.attr == "atomic_add" else -1)],
                keywords=[]), node)
        return node

    def visit_Expr(self, node):
        self.generic_visit(node)
        call = node.value
        if not isinstance(call, ast.Call) or call.keywords:
            return node
        if isinstance(call.func, ast.Name) and call.func.id.startswith("__ti_"):
            return node
        names = [(k, a.id) for k, a in enumerate(call.args) if isinstance(a, ast.Name)]
        if not names:
            return node
        self.counter += 1
        tmp = f"__ti_args{self.counter}"
        fn = f"__ti_fn{self.counter}"
        stmts = [
            ast.Assign(targets=[ast.Name(fn, ast.Store())], value=call.func),
            ast.Assign(targets=[ast.Name(tmp, ast.Store())],
                       value=ast.Call(func=ast.Name("__ti_box_args", ast.Load()),
                                      args=[ast.Name(fn, ast.Load()), ast.List(call.args, ast.Load())], keywords=[])),
            ast.Expr(ast.Call(func=ast.Name(fn, ast.Load()), args=[ast.Starred(ast.Name(tmp, ast.Load()), ast.Load())],
                              keywords=[])),
        ]
        for k, name in names:
            stmts.append(ast.Assign(
                targets=[ast.Name(name, ast.Store())],
                value=ast.Call(func=ast.Name("__ti_after", ast.Load()),
                               args=[ast.Name(name, ast.Load()),
                                     ast.Subscript(ast.Name(tmp, ast.Load()), ast.Constant(k), ast.Load())],
                               keywords=[])))
        return [ast.copy_location(s, node) for s in stmts]


def _ti_all(x):
    return bool(np.all(x))


class _TiCallable:
    """Result of @ti.kernel / @ti.func: compiled lazily on first call."""

    def __init__(self, fn):
        self.fn = fn
        self.compiled = None
        sig = inspect.signature(fn)
        params = [p for p in sig.parameters.values()]
        self.has_self = bool(params) and params[0].name == "self"
        self.template_flags = [isinstance(p.annotation, _Template) for p in params[(1 if self.has_self else 0):]]
        self.__name__ = fn.__name__

    def _compile(self):
        src = textwrap.dedent(inspect.getsource(self.fn))
        tree = ast.parse(src)
        fdef = tree.body[0]
        fdef.decorator_list = []
        for a in fdef.args.args:
            a.annotation = None
        fdef.returns = None
        tree = _Rewriter().visit(tree)
        ast.fix_missing_locations(tree)
        g = dict(self.fn.__globals__)
        g.update(__ti_aug=_aug, __ti_atomic=_atomic, __ti_box_args=_box_args, __ti_after=_after,
                 all=_ti_all, abs=abs)
        code = compile(tree, filename=f"<ti-shim:{self.fn.__qualname__}>", mode="exec")
        exec(code, g)
        self.compiled = g[fdef.name]

    def call(self, *args):
        if self.compiled is None:
            self._compile()
        return self.compiled(*args)

    def __call__(self, *args):
        return self.call(*args)

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return _BoundTi(self, obj)

    @property
    def _ti_wrapper(self):
        return self


class _BoundTi:
    def __init__(self, wrapper, obj):
        self._ti_wrapper = wrapper
        self.obj = obj

    def __call__(self, *args):
        return self._ti_wrapper.call(self.obj, *args)


def kernel(fn):
    return _TiCallable(fn)


def func(fn):
    return _TiCallable(fn)
