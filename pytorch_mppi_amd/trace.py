"""From a plain torch callable to a device functor.

The reference's plugin API is "any Python callable on batched tensors" (/root/reference/src/pytorch_mppi/mppi.py:63-64,
:147-154, :314, :318, :325).  `jit.compile_model` fuses user dynamics when the formulas are given once more as C++
snippets; this module WRITES those snippets: it runs the callable once on symbolic per-sample tensors (batch of one,
every element an expression node), records the arithmetic, and prints it as the `step` / `cost` / `terminal` bodies
`jit.compile_model` compiles around csrc/rollout.hpp.

Symbolic tensors are NOT torch.Tensor subclasses: `SymT` implements the tensor methods users call on states and actions
(indexing, views, arithmetic, reductions, elementwise maths), `__torch_function__` for the torch.* / torch.nn.functional
entry points (torch.cat, torch.clamp, F.linear through an nn.Module, ...) and `__array_ufunc__` for numpy ufuncs applied
to tensors (the reference's own pendulum uses np.sin / np.clip on tensors: tests/pendulum.py:45-46).  Shapes are
concrete (numpy arrays of node ids with a batch axis of size 1), so every view / broadcast / concatenation is numpy's.

Scope: elementwise maths, + - * / ** %, clamp / where / min / max, small constant matrices (captured tensors, nn.Linear
weights), cat / stack / slicing / views, sum / mean / prod over non-batch axes.  Anything else -- data-dependent control
flow, integer indexing by values, in-place writes into the inputs, ops not listed -- raises `TraceUnsupported`, and
the caller keeps the generic path.  The traced functor is VERIFIED before use: compiled for the host (g++) and compared
with the callable on random batches (1e-9 relative, fp64)."""
import ctypes as C
import hashlib
import math
import os
import subprocess
import tempfile

import numpy as np
import torch


class TraceUnsupported(Exception):
    pass


class StaleTrace(Exception):
    """a run-time parameter of a traced functor is no longer what the trace saw (another shape, not a tensor any
    more)"""


class PathParam:
    """A NON-trainable tensor the callables read from a fixed place (watch.Path: an attribute, a closure cell, a global)
    whose values the functor reads from its parameter vector instead of carrying them as constants: the controller
    promotes a tensor to this once it has SEEN it change (`cost.goal = new_goal`; mppi.MPPI._traced_state_moved), so
    that the next
    change is one small copy, not a compile.  `tensor()` is whatever sits at the place now."""
    def __init__(self, path, t):
        self.path, self.shape = path, tuple(t.shape)

    def tensor(self):
        v = self.path.get()
        if not isinstance(v, torch.Tensor) or tuple(v.shape) != self.shape or not v.is_floating_point():
            raise StaleTrace(f"{self.path!r} no longer holds a floating tensor of shape {self.shape}")
        return v


def param_tensor(src):
    """the tensor behind an entry of `param_tensors` (a trainable tensor itself, or what a PathParam's place holds
    now)"""
    return src.tensor() if isinstance(src, PathParam) else src


# ---------------------------------------------------------------------------------------------------------------
# expression graph (hash-consed, constants folded)
# ---------------------------------------------------------------------------------------------------------------
_UNARY = {"neg": lambda a: -a, "sin": math.sin, "cos": math.cos, "tan": math.tan, "tanh": math.tanh, "exp": math.exp,
          "log": math.log, "sqrt": math.sqrt, "abs": abs, "floor": math.floor,
          "sigmoid": lambda a: 1.0 / (1.0 + math.exp(-a)), "sign": lambda a: (a > 0) - (a < 0),
          "erf": math.erf, "atan": math.atan, "asin": math.asin, "acos": math.acos, "sinh": math.sinh,
                  "cosh": math.cosh,
          "expm1": math.expm1, "log1p": math.log1p, "ceil": math.ceil, "round": lambda a: float(np.round(a)),
          "trunc": math.trunc}
_BINARY = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b,
           "min": min, "max": max, "pow": lambda a, b: a ** b, "atan2": math.atan2,
           # (Python's float % is torch.remainder: exact, sign of b)
           "floormod": lambda a, b: a % b, "fmod": math.fmod}
_CMP = {"lt": lambda a, b: a < b, "le": lambda a, b: a <= b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
        "eq": lambda a, b: a == b, "ne": lambda a, b: a != b}


class Graph:
    def __init__(self, max_nodes=60000, device=None, dtype=None, dynamic=()):
        # dynamic: watch.Path places whose (non-trainable) tensors are run-time parameters of this trace (PathParam)
        self.dynamic = {}
        for path in dynamic:
            v = path.get()
            if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 0:
                self.dynamic[id(v)] = (v, path)
        # what the symbolic inputs report as .device / .dtype: the controller's own, so that `net.to(state.device,
        # state.dtype)` or `if state.is_cuda:` inside the callables behave as they will at run time (and a module is not
        # dragged to the host)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.dtype = dtype if dtype is not None else torch.float64
        self.nodes = []          # tuples: ("c", float) | ("x", i) | ("u", n) | ("t",) | ("y", i) | (op, ids...)
        self.index = {}
        # (tensor, version at trace time) of every torch tensor whose VALUES went into constants
        self.captured = []
        # id(tensor made INSIDE the callables from real tensors) -> (tensor, [the tensors it came from]):
        self.derived = {}
        #                          B.to(state.device), W @ W.T, ... are constants of the functor too; the version watch
        #                          must sit on what they were made from (the copy itself is never written again)
        # dense layers kept AS LAYERS (F.linear on a real weight tensor, >= DENSE_MIN multiply-adds): node ("lin",
        # layer, o) is output o of layers[layer] = dict(IN, OUT, inputs=[node ids], wbase, bbase | None) -- weights and
        # bias are parameter- vector reads.  Code generation prints chains of them as mlp_first / mlp_mid / mlp_last
        # calls (csrc/mlp_wide.hpp): fma chains per lane in the ordinary kernels, matrix-core tiles of sixteen samples
        # in the wide kernel
        self.layers = []
        self.dense_layers = os.environ.get("MPPI_TRACE_DENSE", "1") != "0"
        # (tensor, base): TRAINABLE tensors -- element i is the leaf ("p", base + i), read from the
        self.param_tensors = []
        self._param_base = {}    # model's parameter vector at run time (re-gathered when the tensor's version moves)
        self.n_params = 0
        self.max_nodes = max_nodes

    def _mk(self, node):
        i = self.index.get(node)
        if i is None:
            if len(self.nodes) >= self.max_nodes:
                raise TraceUnsupported(f"more than {self.max_nodes} operations per sample")
            i = len(self.nodes)
            self.nodes.append(node)
            self.index[node] = i
        return i

    def const(self, v):
        v = float(v)
        return self._mk(("c", v if v != 0.0 else 0.0))       # -0.0 folded into 0.0

    def leaf(self, kind, i=None):
        return self._mk((kind,) if i is None else (kind, int(i)))

    def param_leaves(self, t, max_params=32768):
        """node ids (shape of t) of the run-time parameter leaves of a trainable tensor (or of a promoted one:
        `dynamic`)"""
        base = self._param_base.get(id(t))
        if base is None:
            if self.n_params + t.numel() > max_params:
                raise TraceUnsupported(f"more than {max_params} trainable parameters")
            base = self.n_params
            self._param_base[id(t)] = base
            dyn = self.dynamic.get(id(t))
            # (the entry keeps t alive, directly or through `dynamic`: id(t) stays unique)
            self.param_tensors.append((PathParam(dyn[1], t) if dyn is not None and dyn[0] is t else t, base))
            self.n_params += t.numel()
        return np.array([self.leaf("p", base + i) for i in range(t.numel())], dtype=np.int64).reshape(tuple(t.shape))

    def param_base(self, t):
        """base of tensor t in the parameter vector (registers it on first use, like param_leaves)"""
        self.param_leaves(t)
        return self._param_base[id(t)]

    def dense(self, inputs, W, b):
        """a layer y = W x + b on node ids `inputs` -> node ids of its OUT outputs"""
        OUT, IN = int(W.shape[0]), int(W.shape[1])
        lid = len(self.layers)
        self.layers.append(dict(IN=IN, OUT=OUT, inputs=[int(i) for i in inputs], wbase=self.param_base(W),
                                bbase=self.param_base(b) if b is not None else None))
        return [self._mk(("lin", lid, o)) for o in range(OUT)]

    def roots_of(self, t):
        d = self.derived.get(id(t))
        return d[1] if d is not None and d[0] is t else [t]

    def note_derived(self, out, srcs):
        roots = []
        for s_ in srcs:
            for r in self.roots_of(s_):
                if not any(r is q for q in roots):
                    roots.append(r)
        if roots and len(self.derived) < 4096 and not any(out is r for r in roots):
            self.derived[id(out)] = (out, roots)               # (holding `out` keeps its id unique)

    def cval(self, i):
        n = self.nodes[i]
        return n[1] if n[0] == "c" else None

    def un(self, op, a):
        ca = self.cval(a)
        if ca is not None:
            try:
                return self.const(_UNARY[op](ca))
            except (ValueError, OverflowError):
                pass
        if op == "neg" and self.nodes[a][0] == "neg":
            return self.nodes[a][1]
        return self._mk((op, a))

    def bin(self, op, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            try:
                return self.const(_BINARY[op](ca, cb))
            except (ValueError, OverflowError, ZeroDivisionError):
                pass
        if op == "add":
            if ca == 0.0:
                return b
            if cb == 0.0:
                return a
        elif op == "sub":
            if cb == 0.0:
                return a
            if ca == 0.0:
                return self.un("neg", b)
        elif op == "mul":
            if ca == 1.0:
                return b
            if cb == 1.0:
                return a
            # (0 * x is NOT folded to 0: x may be inf / nan in torch, and the verification would not see it)
            if ca == -1.0:
                return self.un("neg", b)
            if cb == -1.0:
                return self.un("neg", a)
        elif op == "div":
            if cb == 1.0:
                return a
            if cb is not None and cb != 0.0 and math.isfinite(1.0 / cb) and (1.0 / cb) * cb == 1.0:
                return self.bin("mul", a, self.const(1.0 / cb))      # exact reciprocal (powers of two): same value
        elif op == "pow":
            if cb is not None and cb == int(cb) and 0 <= int(cb) <= 8:
                e = int(cb)
                if e == 0:
                    return self.const(1.0)
                r = a
                # torch.pow with a small integer exponent: repeated products
                for _ in range(e - 1):
                    r = self.bin("mul", r, a)
                return r
            if cb == 0.5:
                return self.un("sqrt", a)
        if op in ("add", "mul", "min", "max") and a > b:
            a, b = b, a                                              # commutative: one node for both orders
        return self._mk((op, a, b))

    def cmp(self, op, a, b):
        return self._mk((op, a, b))

    def table(self, values, idx):
        """("tab", values, idx): element [idx] of a constant table -- a reference / schedule indexed by the timestep"""
        ci = self.cval(idx)
        if ci is not None and ci == int(ci) and 0 <= int(ci) < len(values):
            return self.const(values[int(ci)])
        return self._mk(("tab", tuple(float(v) for v in values), idx))

    def logic(self, op, a, b=None):
        """boolean nodes: ("and" | "or" | "xor", a, b), ("not", a)"""
        if op == "not":
            return self.nodes[a][1] if self.nodes[a][0] == "not" else self._mk(("not", a))
        if a > b:
            a, b = b, a
        return a if a == b and op != "xor" else self._mk((op, a, b))

    def select(self, c, a, b):
        if a == b:
            return a
        return self._mk(("select", c, a, b))


# ---------------------------------------------------------------------------------------------------------------
# symbolic tensor
# ---------------------------------------------------------------------------------------------------------------
def _is_tensor_like(v):
    return isinstance(v, (torch.Tensor, np.ndarray))


class SymT:
    """Per-sample symbolic tensor: `a` is an int64 ndarray of node ids, full shape including the batch axis (size 1)."""
    __array_priority__ = 1000

    def __init__(self, g, a, boolean=False):
        self.g = g
        self.a = np.asarray(a, dtype=np.int64)
        self.boolean = boolean

    # -- construction ------------------------------------------------------------------------------------------
    def _lift(self, v):
        if isinstance(v, SymT):
            return v
        if isinstance(v, SymS):
            return SymT(self.g, np.array(v.i, dtype=np.int64))
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bool:
                raise TraceUnsupported("boolean constant tensors")
            if isinstance(v, torch.nn.Parameter) or v.requires_grad or id(v) in self.g.dynamic:
                # a TRAINABLE tensor: its values are expected to change (online learning of the dynamics, as in the
                # reference's tests/pendulum_approximate.py:47-67,140-170) -- baking them into the functor would go
                # stale with the first optimizer step.  Its elements become reads of the model's parameter vector p[]:
                # the functor stays valid, the vector is re-gathered when the tensor's version counter moves.  (Same for
                # a tensor the controller has seen change at its place: Graph.dynamic, PathParam.)
                return SymT(self.g, self.g.param_leaves(v))
            # captured BY VALUE: the controller watches the version counters of
            for r in self.g.roots_of(v):
                # the tensor -- or of what it was made from inside the callable
                if not any(r is c for c, _ in self.g.captured):
                    self.g.captured.append((r, r._version))
            v = v.detach().cpu().double().numpy()
        if isinstance(v, np.ndarray):
            if v.size > 65536:
                raise TraceUnsupported("constant tensor with more than 65536 elements")
            return SymT(self.g, np.vectorize(self.g.const, otypes=[np.int64])(v.astype(np.float64)))
        if isinstance(v, (int, float, np.floating, np.integer)):
            return SymT(self.g, np.array(self.g.const(float(v)), dtype=np.int64))
        raise TraceUnsupported(f"operand of type {type(v).__name__}")

    def _ew2(self, op, other, reverse=False, cmp=False):
        o = self._lift(other)
        a, b = (o.a, self.a) if reverse else (self.a, o.a)
        try:
            a, b = np.broadcast_arrays(a, b)
        except ValueError as e:
            raise TraceUnsupported(f"broadcast: {e}")
        f = self.g.cmp if cmp else self.g.bin
        out = np.empty(a.shape, dtype=np.int64)
        fa, fb, fo = a.reshape(-1), b.reshape(-1), out.reshape(-1)
        for i in range(fo.size):
            fo[i] = f(op, int(fa[i]), int(fb[i]))
        return SymT(self.g, out, boolean=cmp)

    def _ew1(self, op):
        out = np.empty(self.a.shape, dtype=np.int64)
        fa, fo = self.a.reshape(-1), out.reshape(-1)
        for i in range(fo.size):
            fo[i] = self.g.un(op, int(fa[i]))
        return SymT(self.g, out)

    # -- python protocol ----------------------------------------------------------------------------------------
    def __add__(self, o): return self._ew2("add", o)
    def __radd__(self, o): return self._ew2("add", o, True)
    def __sub__(self, o): return self._ew2("sub", o)
    def __rsub__(self, o): return self._ew2("sub", o, True)
    def __mul__(self, o): return self._ew2("mul", o)
    def __rmul__(self, o): return self._ew2("mul", o, True)
    def __truediv__(self, o): return self._ew2("div", o)
    def __rtruediv__(self, o): return self._ew2("div", o, True)
    def __pow__(self, o): return self._ew2("pow", o)
    def __rpow__(self, o): return self._ew2("pow", o, True)
    def __mod__(self, o): return self._ew2("floormod", o)           # torch's % is Python's: sign of the divisor
    def __neg__(self): return self._ew1("neg")
    def __pos__(self): return self
    def __abs__(self): return self._ew1("abs")
    def __lt__(self, o): return self._ew2("lt", o, cmp=True)
    def __le__(self, o): return self._ew2("le", o, cmp=True)
    def __gt__(self, o): return self._ew2("gt", o, cmp=True)
    def __ge__(self, o): return self._ew2("ge", o, cmp=True)
    def __eq__(self, o): return self._ew2("eq", o, cmp=True)
    def __ne__(self, o): return self._ew2("ne", o, cmp=True)
    __hash__ = object.__hash__
    def eq(self, o): return self == o
    def ne(self, o): return self != o
    def _logic(self, op, o=None):
        if not self.boolean or (o is not None and not (isinstance(o, SymT) and o.boolean)):
            raise TraceUnsupported("logical operator on tensors that are not traced comparisons")
        if o is None:
            return SymT(self.g, np.vectorize(lambda i: self.g.logic("not", int(i)), otypes=[np.int64])(self.a),
                    boolean=True)
        try:
            a, b = np.broadcast_arrays(self.a, o.a)
        except ValueError as e:
            raise TraceUnsupported(f"broadcast: {e}")
        return SymT(self.g, np.vectorize(lambda i, j: self.g.logic(op, int(i), int(j)), otypes=[np.int64])(a, b),
                boolean=True)
    def __and__(self, o): return self._logic("and", o)
    def __or__(self, o): return self._logic("or", o)
    def __xor__(self, o): return self._logic("xor", o)
    def __invert__(self): return self._logic("not")
    __rand__, __ror__ = __and__, __or__
    def logical_and(self, o): return self._logic("and", o)
    def logical_or(self, o): return self._logic("or", o)
    def logical_xor(self, o): return self._logic("xor", o)
    def logical_not(self): return self._logic("not")
    def __matmul__(self, o): return self.matmul(o)
    def __rmatmul__(self, o): return self._lift(o).matmul(self)
    def __bool__(self): raise TraceUnsupported("data-dependent control flow (a tensor used as a Python bool)")
    def __float__(self): raise TraceUnsupported("tensor converted to a Python number")
    __int__ = __index__ = __float__
    def __array__(self, *a, **k): raise TraceUnsupported("tensor converted to a numpy array")
    def __len__(self): return self.a.shape[0]
    def __iter__(self): return (self[i] for i in range(self.a.shape[0]))

    def __getitem__(self, idx):
        if isinstance(idx, SymT) and idx.boolean:
            # x[mask]: a data-dependent selection.  Only the read-modify-write idiom `x[mask] op= scalar` / `x[mask] =
            # ...` can be
            # traced (as a select under the mask): the result is a placeholder that takes scalar arithmetic and goes
            # back into
            # `x[mask] = ...` with the SAME mask
            return _Masked(self.clone(), idx)
        first = idx[0] if isinstance(idx, tuple) else idx
        if isinstance(first, SymT) and first.a.ndim == 0 and not first.boolean:
            # (the timestep after a trip through a torch function)
            first = SymS(self.g, int(first.a))
        if isinstance(first, SymS):
            # table[t] (or table[t, ...]): a constant reference / schedule looked up by the timestep -> one small
            # constant array per selected element in the functor, read at index clamp(t, 0, len - 1)
            rest = idx[1:] if isinstance(idx, tuple) else ()
            cols = np.moveaxis(self.a, 0, -1)                      # (..., N)
            if any(self.g.cval(int(v)) is None for v in cols.reshape(-1)):
                raise TraceUnsupported("indexing a traced (non-constant) tensor by the timestep")
            out = np.empty(cols.shape[:-1], dtype=np.int64)
            for pos in np.ndindex(*out.shape):
                out[pos] = self.g.table([self.g.cval(int(v)) for v in cols[pos]], first.i)
            r = SymT(self.g, out)
            return r[rest] if rest else r
        def chk(i):
            if isinstance(i, (SymT, SymS)):
                raise TraceUnsupported("indexing by a traced value")
            if isinstance(i, torch.Tensor):
                return i.detach().cpu().numpy()
            return i
        idx = tuple(chk(i) for i in idx) if isinstance(idx, tuple) else chk(idx)
        try:
            return SymT(self.g, self.a[idx], self.boolean)
        except (IndexError, TypeError) as e:
            raise TraceUnsupported(f"indexing: {e}")

    def __setitem__(self, idx, v):
        if isinstance(idx, SymT) and idx.boolean:
            if isinstance(v, _Masked):
                if v.mask is not idx and not (v.mask.a.shape == idx.a.shape and np.array_equal(v.mask.a, idx.a)):
                    raise TraceUnsupported("x[mask] = y[other_mask]")
                val = v.full
            else:
                val = self._lift(v)
                if val.a.size != 1:
                    raise TraceUnsupported("x[mask] = tensor (its length depends on the data)")
            try:
                self.a[...] = _where(self.g, idx, val, self).a
            except ValueError as e:
                raise TraceUnsupported(f"masked assignment: {e}")
            return
        if isinstance(v, _Masked):
            raise TraceUnsupported("a masked selection used outside x[mask] = ...")
        v = self._lift(v)
        idx = tuple(i.detach().cpu().numpy() if isinstance(i, torch.Tensor) else i for i in idx) if isinstance(idx,
                tuple) else idx
        try:
            self.a[idx] = v.a          # (the traced inputs are handed to the callable as copies: an in-place write
        except (IndexError, ValueError, TypeError) as e:    # into `state` stays local, like state.clone() first)
            raise TraceUnsupported(f"item assignment: {e}")

    # -- in-place forms: write through self.a (a numpy view of the parent's ids where torch would have a view)
    # --------------
    def _inplace(self, r):
        try:
            self.a[...] = np.broadcast_to(self._lift(r).a, self.a.shape)
        except ValueError as e:
            raise TraceUnsupported(f"in-place operation: {e}")
        return self
    def __iadd__(self, o): return self._inplace(self + o)
    def __isub__(self, o): return self._inplace(self - o)
    def __imul__(self, o): return self._inplace(self * o)
    def __itruediv__(self, o): return self._inplace(self / o)
    def __ipow__(self, o): return self._inplace(self ** o)
    def __imod__(self, o): return self._inplace(self % o)
    def add_(self, o, alpha=1): return self._inplace(self.add(o, alpha=alpha))
    def sub_(self, o, alpha=1): return self._inplace(self.sub(o, alpha=alpha))
    def mul_(self, o): return self._inplace(self * o)
    def div_(self, o): return self._inplace(self / o)
    def pow_(self, o): return self._inplace(self ** o)
    def neg_(self): return self._inplace(-self)
    def abs_(self): return self._inplace(self.abs())
    def clamp_(self, min=None, max=None): return self._inplace(self.clamp(min, max))
    clip_ = clamp_
    def clamp_min_(self, v): return self._inplace(self.clamp(min=v))
    def clamp_max_(self, v): return self._inplace(self.clamp(max=v))
    def copy_(self, o, non_blocking=False): return self._inplace(o)
    def fill_(self, v): return self._inplace(v)
    def zero_(self): return self._inplace(0.0)
    def tanh_(self): return self._inplace(self.tanh())
    def sigmoid_(self): return self._inplace(self.sigmoid())
    def sin_(self): return self._inplace(self.sin())
    def cos_(self): return self._inplace(self.cos())
    def exp_(self): return self._inplace(self.exp())
    def sqrt_(self): return self._inplace(self.sqrt())
    def remainder_(self, o): return self._inplace(self % o)
    def fmod_(self, o): return self._inplace(self.fmod(o))
    def masked_fill(self, mask, value): return _where(self.g, mask, _as_sym(self.g, value), self)
    def masked_fill_(self, mask, value): return self._inplace(self.masked_fill(mask, value))

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != "__call__" or kw.get("out") is not None:
            raise TraceUnsupported(f"numpy {ufunc.__name__}.{method}")
        name = _NP_UFUNCS.get(ufunc.__name__)
        if name is None:
            raise TraceUnsupported(f"numpy ufunc {ufunc.__name__}")
        return _call(self.g, name, inputs, {})

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        g = _graph_of(args, kwargs)
        if name in ("__get__",):          # attribute descriptors (Tensor.T, .shape, ...) reach here for some builds
            raise TraceUnsupported(f"torch attribute {func}")
        return _call(g, name, args, kwargs)

    # -- attributes ---------------------------------------------------------------------------------------------
    @property
    def shape(self): return torch.Size(self.a.shape)
    @property
    def ndim(self): return self.a.ndim
    @property
    def dtype(self): return self.g.dtype
    @property
    def device(self): return self.g.device
    @property
    def T(self): return SymT(self.g, self.a.T)
    @property
    def mT(self): return SymT(self.g, np.swapaxes(self.a, -1, -2))
    @property
    def requires_grad(self): return False
    @property
    def is_cuda(self): return self.g.device.type == "cuda"
    def size(self, d=None): return self.shape if d is None else self.a.shape[d]
    def dim(self): return self.a.ndim
    ndimension = dim
    def nelement(self): return self.a.size
    def is_floating_point(self): return not self.boolean
    def is_contiguous(self, *a, **k): return True
    @property
    def data(self): return self
    @property
    def grad_fn(self): return None
    def numpy(self, *a, **k): raise TraceUnsupported("tensor converted to a numpy array")
    def tolist(self): raise TraceUnsupported("tensor converted to a Python list")
    def item(self): raise TraceUnsupported("tensor converted to a Python number (.item())")
    def type(self, *a, **k): return self if a or k else "torch.DoubleTensor"
    def _new(self, shape, v): return SymT(self.g, np.full(self._shape_args(shape), self.g.const(v), dtype=np.int64))
    def new_zeros(self, *shape, **k): return self._new(shape, 0.0)
    def new_ones(self, *shape, **k): return self._new(shape, 1.0)
    def new_empty(self, *shape, **k): return self._new(shape, 0.0)
    def new_full(self, shape, fill_value, **k): return self._new((shape,), float(fill_value))
    def new_tensor(self, data, **k): return self._lift(torch.as_tensor(data, dtype=torch.float64))
    def numel(self): return self.a.size
    def t(self): return self.T

    # -- no-ops / views -----------------------------------------------------------------------------------------
    def clone(self, *a, **k): return SymT(self.g, self.a.copy(), self.boolean)
    def contiguous(self, *a, **k): return self
    def detach(self): return self
    def to(self, *a, **k): return self
    def type_as(self, o): return self
    def float(self): return self
    def double(self): return self
    def cpu(self): return self
    def cuda(self, *a, **k): return self
    def requires_grad_(self, *a, **k): return self

    def _shape_args(self, s):
        if len(s) == 1 and isinstance(s[0], (tuple, list, torch.Size)):
            s = tuple(s[0])
        return tuple(int(v) for v in s)

    def view(self, *s):
        try:
            return SymT(self.g, self.a.reshape(self._shape_args(s)), self.boolean)
        except ValueError as e:
            raise TraceUnsupported(f"view: {e}")
    reshape = view

    def view_as(self, o): return self.view(*o.shape)
    def reshape_as(self, o): return self.view(*o.shape)
    def flatten(self, start_dim=0, end_dim=-1):
        sh = list(self.a.shape)
        e = end_dim % len(sh)
        s = start_dim % len(sh)
        return self.view(*(sh[:s] + [-1] + sh[e + 1:]))
    def unsqueeze(self, d): return SymT(self.g, np.expand_dims(self.a, d if d >= 0 else d + self.a.ndim + 1),
            self.boolean)
    def squeeze(self, d=None):
        if d is None:
            return SymT(self.g, np.squeeze(self.a), self.boolean)
        return SymT(self.g, np.squeeze(self.a, d), self.boolean) if self.a.shape[d] == 1 else self
    def expand(self, *s):
        s = self._shape_args(s)
        s = tuple(self.a.shape[i - (len(s) - self.a.ndim)] if v == -1 else v for i, v in enumerate(s))
        try:
            return SymT(self.g, np.broadcast_to(self.a, s), self.boolean)
        except ValueError as e:
            raise TraceUnsupported(f"expand: {e}")
    def expand_as(self, o): return self.expand(*o.shape)
    def repeat(self, *s): return SymT(self.g, np.tile(self.a, self._shape_args(s)), self.boolean)
    def transpose(self, d0, d1): return SymT(self.g, np.swapaxes(self.a, d0, d1), self.boolean)
    swapaxes = swapdims = transpose
    def movedim(self, src, dst): return SymT(self.g, np.moveaxis(self.a, src, dst), self.boolean)
    moveaxis = movedim
    def tile(self, *s): return self.repeat(*s)
    def permute(self, *d): return SymT(self.g, np.transpose(self.a, self._shape_args(d)), self.boolean)
    def unbind(self, dim=0): return tuple(SymT(self.g, np.take(self.a, i, axis=dim)) for i in range(self.a.shape[dim]))
    def chunk(self, n, dim=0): return tuple(SymT(self.g, p) for p in np.array_split(self.a, n, axis=dim))
    def split(self, size, dim=0):
        if isinstance(size, int):
            cuts = list(range(size, self.a.shape[dim], size))
        else:
            cuts = list(np.cumsum(size)[:-1])
        return tuple(SymT(self.g, p) for p in np.split(self.a, cuts, axis=dim))
    def narrow(self, dim, start, length):
        sl = [slice(None)] * self.a.ndim
        sl[dim] = slice(start, start + length)
        return SymT(self.g, self.a[tuple(sl)])
    def select(self, dim, index): return SymT(self.g, np.take(self.a, index, axis=dim))

    # -- elementwise --------------------------------------------------------------------------------------------
    def sin(self): return self._ew1("sin")
    def cos(self): return self._ew1("cos")
    def tan(self): return self._ew1("tan")
    def tanh(self): return self._ew1("tanh")
    def exp(self): return self._ew1("exp")
    def log(self): return self._ew1("log")
    def sqrt(self): return self._ew1("sqrt")
    def abs(self): return self._ew1("abs")
    def neg(self): return self._ew1("neg")
    def floor(self): return self._ew1("floor")
    def sign(self): return self._ew1("sign")
    def sigmoid(self): return self._ew1("sigmoid")
    def relu(self, inplace=False):
        r = self._ew2("max", 0.0)
        return self._inplace(r) if inplace else r
    def relu_(self): return self.relu(True)
    def square(self): return self._ew2("mul", self)
    def erf(self): return self._ew1("erf")
    def atan(self): return self._ew1("atan")
    arctan = atan
    def asin(self): return self._ew1("asin")
    arcsin = asin
    def acos(self): return self._ew1("acos")
    arccos = acos
    def sinh(self): return self._ew1("sinh")
    def cosh(self): return self._ew1("cosh")
    def expm1(self): return self._ew1("expm1")
    def log1p(self): return self._ew1("log1p")
    def log2(self): return self.log() * (1.0 / math.log(2.0))
    def log10(self): return self.log() * (1.0 / math.log(10.0))
    def exp2(self): return (self * math.log(2.0)).exp()
    def ceil(self): return self._ew1("ceil")
    def round(self, decimals=0):
        if decimals != 0:
            raise TraceUnsupported("round(decimals != 0)")
        return self._ew1("round")
    def trunc(self): return self._ew1("trunc")
    fix = trunc
    def frac(self): return self - self.trunc()
    def lerp(self, end, weight): return self + (self._lift(end) - self) * weight
    def addcmul(self, t1, t2, value=1): return self + (self._lift(t1) * t2) * value
    def addcdiv(self, t1, t2, value=1): return self + (self._lift(t1) / t2) * value
    def hypot(self, o): return (self * self + self._lift(o) * o).sqrt()
    def logaddexp(self, o):
        o = self._lift(o)
        m = self.maximum(o)
        return m + ((self - m).exp() + (o - m).exp()).log()
    def flip(self, dims=None, *more):
        dims = (dims,) + more if isinstance(dims, int) else tuple(dims)
        return SymT(self.g, np.flip(self.a, axis=dims), self.boolean)
    def roll(self, shifts, dims=None):
        return SymT(self.g, np.roll(self.a, shifts, axis=dims), self.boolean)
    def cumsum(self, dim, dtype=None):
        a = np.moveaxis(self.a, dim, 0).copy()
        for r in range(1, a.shape[0]):
            fo, fp = a[r].reshape(-1), a[r - 1].reshape(-1)
            a[r] = np.array([self.g.bin("add", int(p_), int(o_)) for p_, o_ in zip(fp, fo)],
                    dtype=np.int64).reshape(a[r].shape)
        return SymT(self.g, np.moveaxis(a, 0, dim))
    def cumprod(self, dim, dtype=None):
        a = np.moveaxis(self.a, dim, 0).copy()
        for r in range(1, a.shape[0]):
            fo, fp = a[r].reshape(-1), a[r - 1].reshape(-1)
            a[r] = np.array([self.g.bin("mul", int(p_), int(o_)) for p_, o_ in zip(fp, fo)],
                    dtype=np.int64).reshape(a[r].shape)
        return SymT(self.g, np.moveaxis(a, 0, dim))
    def outer(self, o):
        o = self._lift(o)
        return self.unsqueeze(-1) * o.unsqueeze(-2)
    def diagonal(self, offset=0, dim1=0, dim2=1): return SymT(self.g, np.diagonal(self.a, offset, dim1, dim2),
            self.boolean)
    def trace(self): return self.diagonal().sum(-1)
    def diag(self, diagonal=0):
        if self.a.ndim == 2:
            return SymT(self.g, np.diagonal(self.a, diagonal), self.boolean)
        if self.a.ndim == 1 and diagonal == 0:
            out = np.full((self.a.size, self.a.size), self.g.const(0.0), dtype=np.int64)
            out[np.arange(self.a.size), np.arange(self.a.size)] = self.a
            return SymT(self.g, out)
        raise TraceUnsupported("diag of this shape")
    def tril(self, diagonal=0):
        z = self.g.const(0.0)
        m = np.tril(np.ones(self.a.shape[-2:], dtype=bool), diagonal)
        return SymT(self.g, np.where(m, self.a, z))
    def triu(self, diagonal=0):
        z = self.g.const(0.0)
        m = np.triu(np.ones(self.a.shape[-2:], dtype=bool), diagonal)
        return SymT(self.g, np.where(m, self.a, z))
    def cross(self, o, dim=-1):
        o = self._lift(o)
        a = [SymT(self.g, np.take(self.a, i, axis=dim)) for i in range(3)]
        b = [SymT(self.g, np.take(np.broadcast_to(o.a, self.a.shape), i, axis=dim)) for i in range(3)]
        c = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
        return SymT(self.g, np.stack([v.a for v in c], axis=dim))
    def reciprocal(self): return SymT(self.g, np.array(self.g.const(1.0)))._ew2("div", self)
    def rsqrt(self): return self.sqrt().reciprocal()
    def add(self, o, alpha=1): return self + (o if alpha == 1 else o * alpha)
    def sub(self, o, alpha=1): return self - (o if alpha == 1 else o * alpha)
    def mul(self, o): return self * o
    def div(self, o): return self / o
    def pow(self, o): return self ** o
    def remainder(self, o): return self % o
    def fmod(self, o): return self._ew2("fmod", o)
    def atan2(self, o): return self._ew2("atan2", o)
    def maximum(self, o): return self._ew2("max", o)
    def minimum(self, o): return self._ew2("min", o)
    fmax, fmin = maximum, minimum
    def mv(self, o): return self.matmul(o)
    def inner(self, o): return (self * o).sum(-1) if self.a.ndim == 1 else self.matmul(self._lift(o).mT
            if self._lift(o).a.ndim > 1 else o)
    def any(self, dim=None, keepdim=False):
        if not self.boolean:
            raise TraceUnsupported("any() of a tensor that is not a traced comparison")
        r, _ = self._reduce_with(lambda i, j: self.g.logic("or", i, j), dim, keepdim)
        r.boolean = True
        return r
    def all(self, dim=None, keepdim=False):
        if not self.boolean:
            raise TraceUnsupported("all() of a tensor that is not a traced comparison")
        r, _ = self._reduce_with(lambda i, j: self.g.logic("and", i, j), dim, keepdim)
        r.boolean = True
        return r
    def lt(self, o): return self < o
    def le(self, o): return self <= o
    def gt(self, o): return self > o
    def ge(self, o): return self >= o

    def clamp(self, min=None, max=None, out=None, **kw):
        if out is not None or kw:
            raise TraceUnsupported("clamp(out=...)")
        r = self
        if min is not None and max is not None and not isinstance(min, (SymT, torch.Tensor)) and not isinstance(max,
                (SymT, torch.Tensor)):
            lo, hi = self._lift(min), self._lift(max)
            out = np.empty(self.a.shape, dtype=np.int64)
            fa, fo = self.a.reshape(-1), out.reshape(-1)
            for i in range(fo.size):
                fo[i] = self.g._mk(("clamp", int(fa[i]), int(lo.a), int(hi.a)))
            return SymT(self.g, out)
        if min is not None:
            r = r._ew2("max", min)
        if max is not None:
            r = r._ew2("min", max)
        return r
    clip = clamp
    def clamp_min(self, v): return self.clamp(min=v)
    def clamp_max(self, v): return self.clamp(max=v)

    def where(self, cond, other):          # Tensor.where(condition, other): self where cond else other
        return _where(self.g, cond, self, other)

    # -- reductions ---------------------------------------------------------------------------------------------
    def _reduce(self, op, dim, keepdim):
        return self._reduce_with(lambda i, j: self.g.bin(op, i, j), dim, keepdim)

    def _reduce_with(self, f, dim, keepdim):
        a = self.a
        if dim is None:
            dims = tuple(range(a.ndim))
        else:
            dims = tuple(d % a.ndim for d in (dim if isinstance(dim, (tuple, list)) else (dim,)))
        moved = np.moveaxis(a, dims, tuple(range(len(dims))))
        flat = moved.reshape((-1,) + moved.shape[len(dims):])
        out = flat[0].copy()
        for r in range(1, flat.shape[0]):                       # index order, like a sequential sum
            fo, fr = out.reshape(-1), flat[r].reshape(-1)
            for i in range(fo.size):
                fo[i] = f(int(fo[i]), int(fr[i]))
        if keepdim:
            for d in sorted(dims):
                out = np.expand_dims(out, d)
        return SymT(self.g, out), flat.shape[0]

    def sum(self, dim=None, keepdim=False, dtype=None):
        return self._reduce("add", dim, keepdim)[0]
    def prod(self, dim=None, keepdim=False):
        return self._reduce("mul", dim, keepdim)[0]
    def mean(self, dim=None, keepdim=False):
        r, n = self._reduce("add", dim, keepdim)
        return r / float(n)
    def amax(self, dim=None, keepdim=False): return self._reduce("max", dim, keepdim)[0]
    def amin(self, dim=None, keepdim=False): return self._reduce("min", dim, keepdim)[0]
    def max(self, dim=None, keepdim=False):
        """max() -> tensor; max(dim) -> (values, indices): only `.values` / [0] can be traced"""
        if dim is None:
            return self.amax()
        if isinstance(dim, (SymT, torch.Tensor)):
            return self.maximum(dim)
        return _ValuesOnly(self.amax(dim, keepdim))
    def min(self, dim=None, keepdim=False):
        if dim is None:
            return self.amin()
        if isinstance(dim, (SymT, torch.Tensor)):
            return self.minimum(dim)
        return _ValuesOnly(self.amin(dim, keepdim))
    def var(self, dim=None, unbiased=True, keepdim=False, correction=None):
        if isinstance(dim, bool):                                   # var(unbiased)
            dim, unbiased = None, dim
        corr = (1 if unbiased else 0) if correction is None else correction
        mean, n = self._reduce("add", dim, True)
        d = self - mean / float(n)
        return (d * d).sum(dim, keepdim) / float(n - corr)
    def std(self, dim=None, unbiased=True, keepdim=False, correction=None):
        return self.var(dim, unbiased, keepdim, correction).sqrt()
    def norm(self, p=2, dim=None, keepdim=False):
        if p in (2, 2.0, "fro", None):
            return (self * self).sum(dim, keepdim).sqrt()
        if p in (1, 1.0):
            return self.abs().sum(dim, keepdim)
        if p == float("inf"):
            return self.abs().amax(dim, keepdim)
        if isinstance(p, (int, float)) and p > 0:
            return (self.abs() ** float(p)).sum(dim, keepdim) ** (1.0 / float(p))
        raise TraceUnsupported(f"norm with p = {p}")

    def matmul(self, o):
        o = self._lift(o)
        a, b = self.a, o.a
        if a.ndim == 0 or b.ndim == 0:
            raise TraceUnsupported("matmul with a 0-d operand")
        a2 = a if a.ndim > 1 else a[None, :]
        b2 = b if b.ndim > 1 else b[:, None]
        if a2.shape[-1] != b2.shape[-2]:
            raise TraceUnsupported(f"matmul shapes {a.shape} @ {b.shape}")
        try:
            batch = np.broadcast_shapes(a2.shape[:-2], b2.shape[:-2])
        except ValueError as e:
            raise TraceUnsupported(f"matmul: {e}")
        a2 = np.broadcast_to(a2, batch + a2.shape[-2:])
        b2 = np.broadcast_to(b2, batch + b2.shape[-2:])
        out = np.empty(batch + (a2.shape[-2], b2.shape[-1]), dtype=np.int64)
        g = self.g
        for bi in np.ndindex(*batch):
            for i in range(a2.shape[-2]):
                for j in range(b2.shape[-1]):
                    acc = None
                    for k in range(a2.shape[-1]):
                        ia, ib = int(a2[bi + (i, k)]), int(b2[bi + (k, j)])
                        if g.cval(ia) == 0.0 or g.cval(ib) == 0.0:
                            # a structural zero of a CONSTANT matrix: torch adds 0 * x = 0 for finite x; dropping the
                            # term is exact for finite states (the sparse B / selection matrices of test code)
                            continue
                        p = g.bin("mul", ia, ib)
                        acc = p if acc is None else g.bin("add", acc, p)
                    out[bi + (i, j)] = g.const(0.0) if acc is None else acc
        if a.ndim == 1:
            out = out[..., 0, :]
        if b.ndim == 1:
            out = out[..., 0]
        return SymT(self.g, out)
    mm = matmul
    bmm = matmul
    def dot(self, o): return (self * o).sum()


class _Masked:
    """`x[mask]` of a traced boolean mask: the full-shape values with the mask beside them.  Scalar arithmetic only; it
    can
    go back into `x[mask] = ...` (see SymT.__setitem__); any other use is refused."""
    def __init__(self, full, mask):
        self.full, self.mask = full, mask

    def _b(self, op, o, rev=False):
        if isinstance(o, (SymT, SymS, _Masked)) or (isinstance(o, (torch.Tensor, np.ndarray)) and o.size != 1
                if isinstance(o, np.ndarray) else
                                                      isinstance(o, torch.Tensor) and o.numel() != 1):
            raise TraceUnsupported("arithmetic between a masked selection and a tensor")
        return _Masked(self.full._ew2(op, o, reverse=rev), self.mask)
    def __add__(self, o): return self._b("add", o)
    def __radd__(self, o): return self._b("add", o, True)
    def __sub__(self, o): return self._b("sub", o)
    def __rsub__(self, o): return self._b("sub", o, True)
    def __mul__(self, o): return self._b("mul", o)
    def __rmul__(self, o): return self._b("mul", o, True)
    def __truediv__(self, o): return self._b("div", o)
    def __mod__(self, o): return self._b("floormod", o)
    def __neg__(self): return _Masked(-self.full, self.mask)
    def __getattr__(self, name):
        raise TraceUnsupported(f"a masked selection x[mask] used as a tensor (.{name}): its length depends on the data")


class _ValuesOnly:
    """result of Tensor.max(dim) / min(dim): the values can be traced, the indices cannot (they would be
    data-dependent)"""
    def __init__(self, values): self.values = values
    @property
    def indices(self): raise TraceUnsupported("argmax / argmin indices")
    def __getitem__(self, i):
        if i == 0:
            return self.values
        raise TraceUnsupported("argmax / argmin indices")
    def __iter__(self): raise TraceUnsupported("argmax / argmin indices (unpacking values, indices)")


class SymS:
    """Symbolic Python scalar (the timestep `t` of step-dependent callables): arithmetic only."""
    def __init__(self, g, i):
        self.g, self.i = g, i

    def _b(self, op, o, rev=False):
        if isinstance(o, SymT):
            return o._ew2(op, self, reverse=not rev)
        if isinstance(o, SymS):
            oi = o.i
        elif isinstance(o, (int, float)):
            oi = self.g.const(o)
        elif isinstance(o, torch.Tensor):
            return SymT(self.g, np.array(self.i))._ew2(op, o, reverse=rev)
        else:
            return NotImplemented
        return SymS(self.g, self.g.bin(op, oi, self.i) if rev else self.g.bin(op, self.i, oi))

    def __add__(self, o): return self._b("add", o)
    def __radd__(self, o): return self._b("add", o, True)
    def __sub__(self, o): return self._b("sub", o)
    def __rsub__(self, o): return self._b("sub", o, True)
    def __mul__(self, o): return self._b("mul", o)
    def __rmul__(self, o): return self._b("mul", o, True)
    def __truediv__(self, o): return self._b("div", o)
    def __rtruediv__(self, o): return self._b("div", o, True)
    def __pow__(self, o): return self._b("pow", o)
    def __mod__(self, o): return self._b("floormod", o)
    def __floordiv__(self, o):
        r = self._b("div", o)
        return SymS(self.g, self.g.un("floor", r.i)) if isinstance(r, SymS) else r.floor()
    def __neg__(self): return SymS(self.g, self.g.un("neg", self.i))
    def __bool__(self): raise TraceUnsupported("control flow on the timestep")
    def __index__(self): raise TraceUnsupported("indexing by the timestep")
    __int__ = __index__
    def __float__(self): raise TraceUnsupported("the timestep converted to a Python float")
    def __lt__(self, o): raise TraceUnsupported("comparison on the timestep")
    # == / != / hashing must fail as loudly as < does: left at the object defaults, `if t == T - 1:` would evaluate to a
    # plain False while tracing and the branch would be dropped without a word (`t in (...)`, dict lookups by t: the
    # same)
    __le__ = __gt__ = __ge__ = __eq__ = __ne__ = __lt__
    def __hash__(self): raise TraceUnsupported("the timestep used as a dictionary key / set member")
    __array_priority__ = 1000
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = tuple(SymT(a.g, np.array(a.i)) if isinstance(a, SymS) else a for a in args)
        return _call(_graph_of(args, kwargs or {}), getattr(func, "__name__", str(func)), args, kwargs or {})


_DUNDERS = {"__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__",
        "__pow__",
            "__rpow__", "__mod__", "__neg__", "__pos__", "__abs__", "__lt__", "__le__", "__gt__", "__ge__",
                    "__matmul__",
            "__rmatmul__", "__getitem__"}
_NP_UFUNCS = {"sin": "sin", "cos": "cos", "tan": "tan", "tanh": "tanh", "exp": "exp", "log": "log", "sqrt": "sqrt",
              "absolute": "abs", "fabs": "abs", "negative": "neg", "square": "square", "add": "add", "subtract": "sub",
              "multiply": "mul", "true_divide": "div", "divide": "div", "power": "pow", "maximum": "maximum",
              "minimum": "minimum", "floor": "floor", "sign": "sign", "arctan2": "atan2", "remainder": "remainder",
              "mod": "remainder", "fmod": "fmod", "clip": "clamp", "less": "lt", "greater": "gt", "less_equal": "le",
              "greater_equal": "ge"}


def _graph_of(args, kwargs):
    def walk(v):
        if isinstance(v, (SymT, SymS)):
            return v.g
        if isinstance(v, (tuple, list)):
            for e in v:
                g = walk(e)
                if g is not None:
                    return g
        return None
    for v in list(args) + list(kwargs.values()):
        g = walk(v)
        if g is not None:
            return g
    raise TraceUnsupported("no traced operand")


def _as_sym(g, v):
    if isinstance(v, SymT):
        return v
    return SymT(g, np.array(g.const(0.0)))._lift(v)


def _where(g, cond, a, b):
    if not isinstance(cond, SymT) or not cond.boolean:
        raise TraceUnsupported("where() on a condition that does not come from a traced comparison")
    a, b = _as_sym(g, a), _as_sym(g, b)
    try:
        c_, a_, b_ = np.broadcast_arrays(cond.a, a.a, b.a)
    except ValueError as e:
        raise TraceUnsupported(f"where: {e}")
    out = np.empty(c_.shape, dtype=np.int64)
    fc, fa, fb, fo = c_.reshape(-1), a_.reshape(-1), b_.reshape(-1), out.reshape(-1)
    for i in range(fo.size):
        fo[i] = g.select(int(fc[i]), int(fa[i]), int(fb[i]))
    return SymT(g, out)


def _einsum(g, eq, ops):
    """explicit-output einsum ("bi,ij->bj", "bi,ij,bj->b", ...) as sums of products, terms in index order"""
    if not isinstance(eq, str):
        raise TraceUnsupported("einsum in the sublist format")
    eq = eq.replace(" ", "")
    if "->" not in eq or "." in eq:
        raise TraceUnsupported("einsum without an explicit output / with an ellipsis")
    lhs, out = eq.split("->")
    terms = lhs.split(",")
    syms = [_as_sym(g, o) for o in ops]
    if len(terms) != len(syms):
        raise TraceUnsupported("einsum: operand count")
    sizes = {}
    for t, sy in zip(terms, syms):
        if len(t) != sy.a.ndim:
            raise TraceUnsupported(f"einsum: '{t}' against a {sy.a.ndim}-d operand")
        for ch, n in zip(t, sy.a.shape):
            if sizes.setdefault(ch, n) != n:
                raise TraceUnsupported(f"einsum: size of index '{ch}'")
    if any(c not in sizes for c in out) or len(set(out)) != len(out):
        raise TraceUnsupported("einsum: output indices")
    summed = [c for c in sizes if c not in out]
    res = np.empty([sizes[c] for c in out], dtype=np.int64)
    for oi in np.ndindex(*res.shape):
        env = dict(zip(out, oi))
        acc = None
        for si in np.ndindex(*[sizes[c] for c in summed]):
            env.update(zip(summed, si))
            prod = None
            for t, sy in zip(terms, syms):
                node = int(sy.a[tuple(env[c] for c in t)])
                prod = node if prod is None else g.bin("mul", prod, node)
            if g.cval(prod) == 0.0:
                continue                               # structural zero of constant operands (see matmul)
            acc = prod if acc is None else g.bin("add", acc, prod)
        res[oi] = g.const(0.0) if acc is None else acc
    return SymT(g, res)


def _call(g, name, args, kwargs):
    """torch.* / torch.nn.functional.* / Tensor.* entry points by name."""
    name = {"_threshold": "threshold"}.get(name, name)            # (F.threshold is the private function _threshold)
    a0 = _as_sym(g, args[0]) if args and not isinstance(args[0], (tuple, list, str)) else None
    rest = args[1:]
    if name in ("cat", "concatenate", "concat", "stack", "hstack", "vstack"):
        seq = [_as_sym(g, v).a for v in args[0]]
        dim = kwargs.get("dim", rest[0] if rest else 0)
        try:
            if name == "stack":
                return SymT(g, np.stack(seq, axis=dim))
            if name == "hstack":
                return SymT(g, np.hstack(seq))
            if name == "vstack":
                return SymT(g, np.vstack(seq))
            return SymT(g, np.concatenate(seq, axis=dim))
        except ValueError as e:
            raise TraceUnsupported(f"{name}: {e}")
    if name == "where":
        if len(args) != 3:
            raise TraceUnsupported("where(condition) without values")
        return _where(g, args[0], args[1], args[2])
    if name in ("zeros_like", "ones_like", "full_like", "empty_like"):
        v = {"zeros_like": 0.0, "ones_like": 1.0, "empty_like": 0.0}.get(name, rest[0] if rest
                else kwargs.get("fill_value"))
        return SymT(g, np.full(a0.a.shape, g.const(v), dtype=np.int64))
    if name == "linear":                               # F.linear(input, weight, bias): nn.Linear inside a module
        b_ = rest[1] if len(rest) > 1 else kwargs.get("bias")
        W_ = rest[0]
        if (g.dense_layers and isinstance(W_, torch.Tensor) and W_.dim() == 2 and W_.is_floating_point()
                and W_.numel() >= DENSE_MIN
                and isinstance(a0, SymT) and a0.a.ndim >= 1 and a0.a.shape[-1] == W_.shape[1] and not a0.boolean
                and (b_ is None or (isinstance(b_, torch.Tensor) and b_.dim() == 1 and b_.numel() == W_.shape[0]))
                and id(W_) not in g.dynamic and (b_ is None or id(b_) not in g.dynamic)):
            # a dense layer stays a layer: its weights become parameter-vector reads (trainable or not: a frozen
            # network's weights are followed by version counter and storage like any parameter), its outputs `lin` nodes
            rows = a0.a.reshape(-1, a0.a.shape[-1])
            out = np.empty((rows.shape[0], int(W_.shape[0])), dtype=np.int64)
            for r in range(rows.shape[0]):
                out[r] = g.dense(rows[r], W_, b_)
            return SymT(g, out.reshape(a0.a.shape[:-1] + (int(W_.shape[0]),)))
        w = _as_sym(g, rest[0])
        out = a0.matmul(w.T)
        b = rest[1] if len(rest) > 1 else kwargs.get("bias")
        return out + b if b is not None else out
    if name == "einsum":
        ops = args[1] if len(args) == 2 and isinstance(args[1], (tuple, list)) else args[1:]
        return _einsum(g, args[0], ops)
    if name in ("index_select", "gather", "scatter", "nonzero", "argmax", "argmin", "sort", "topk", "argsort"):
        raise TraceUnsupported(f"torch.{name}")
    if name in ("max", "min"):
        return getattr(a0, name)(*rest, **kwargs)
    if name in ("linalg_norm", "linalg_vector_norm", "vector_norm", "norm"):
        p_ = kwargs.get("ord", kwargs.get("p", rest[0] if rest else 2))
        dim = kwargs.get("dim", rest[1] if len(rest) > 1 else None)
        return a0.norm(2 if p_ is None else p_, dim, kwargs.get("keepdim", rest[2] if len(rest) > 2 else False))
    # `if not torch.is_tensor(x): x = torch.tensor(x)` on a traced input
    if name in ("tensor", "as_tensor", "asarray"):
        return a0
    if name == "cdist":                                # torch.cdist(x1 (..,P,M), x2 (..,R,M), p=2) -> (..,P,R)
        x2 = _as_sym(g, rest[0])
        p_ = float(kwargs.get("p", rest[1] if len(rest) > 1 else 2.0))
        d = a0.unsqueeze(-2) - x2.unsqueeze(-3)
        return d.norm(p_ if p_ != 2.0 else 2, -1)
    if name == "normalize":                            # F.normalize(input, p=2, dim=1, eps=1e-12)
        p_ = kwargs.get("p", rest[0] if rest else 2.0)
        dim = kwargs.get("dim", rest[1] if len(rest) > 1 else 1)
        eps = kwargs.get("eps", rest[2] if len(rest) > 2 else 1e-12)
        return a0 / a0.norm(p_, dim, True).clamp(min=eps)
    if name in ("addmm", "addmv", "baddbmm", "addbmm"):
        if name == "addbmm":
            raise TraceUnsupported("torch.addbmm")
        prod = _as_sym(g, rest[0]).matmul(rest[1])
        alpha, beta = kwargs.get("alpha", 1), kwargs.get("beta", 1)
        return (a0 if beta == 1 else a0 * beta) + (prod if alpha == 1 else prod * alpha)
    if name in ("mse_loss", "l1_loss", "smooth_l1_loss", "huber_loss"):
        d = a0 - rest[0]
        if name == "mse_loss":
            e = d * d
        elif name == "l1_loss":
            e = d.abs()
        else:
            beta = kwargs.get("beta", 1.0) if name == "smooth_l1_loss" else kwargs.get("delta", 1.0)
            ad = d.abs()
            quad = d * d * (0.5 / beta) if name == "smooth_l1_loss" else d * d * 0.5
            lin = ad - 0.5 * beta if name == "smooth_l1_loss" else (ad - 0.5 * beta) * beta
            e = _where(g, ad < beta, quad, lin)
        red = kwargs.get("reduction", "mean")
        return e if red == "none" else (e.sum() if red == "sum" else e.mean())
    if name in ("softmax", "log_softmax", "softmin"):
        dim = kwargs.get("dim", rest[0] if rest else None)
        if dim is None:
            raise TraceUnsupported(f"{name} without dim")
        x = -a0 if name == "softmin" else a0
        sh = x - x.amax(dim, True)
        if name == "log_softmax":
            return sh - sh.exp().sum(dim, True).log()
        e = sh.exp()
        return e / e.sum(dim, True)
    if name == "layer_norm":                           # F.layer_norm(input, normalized_shape, weight, bias, eps)
        nshape = tuple(rest[0]) if not isinstance(rest[0], int) else (rest[0],)
        w = kwargs.get("weight", rest[1] if len(rest) > 1 else None)
        b = kwargs.get("bias", rest[2] if len(rest) > 2 else None)
        eps = kwargs.get("eps", rest[3] if len(rest) > 3 else 1e-5)
        dims = tuple(range(a0.a.ndim - len(nshape), a0.a.ndim))
        mu = a0.mean(dims, True)
        d = a0 - mu
        y = d / ((d * d).mean(dims, True) + eps).sqrt()
        if w is not None:
            y = y * w
        return y + b if b is not None else y
    if name in ("hardtanh", "relu6", "elu", "selu", "celu", "gelu", "tanhshrink", "softsign", "mish", "hardswish",
                "hardsigmoid", "logsigmoid", "log_sigmoid", "threshold", "softshrink", "hardshrink", "silu", "relu",
                        "relu_", "elu_",
                "hardtanh_", "threshold_"):
        name = {"log_sigmoid": "logsigmoid"}.get(name, name.rstrip("_"))
        if name == "relu":
            return a0.relu()
        if name == "silu":
            return a0 * a0.sigmoid()
        if name in ("hardtanh", "relu6"):
            lo = 0.0 if name == "relu6" else kwargs.get("min_val", rest[0] if rest else -1.0)
            hi = 6.0 if name == "relu6" else kwargs.get("max_val", rest[1] if len(rest) > 1 else 1.0)
            return a0.clamp(float(lo), float(hi))
        if name in ("elu", "celu"):
            alpha = float(kwargs.get("alpha", rest[0] if rest else 1.0))
            neg = (a0.expm1() if name == "elu" else (a0 / alpha).expm1()) * alpha
            return _where(g, a0 > 0.0, a0, neg)
        if name == "selu":
            alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
            return _where(g, a0 > 0.0, a0, a0.expm1() * alpha) * scale
        if name == "gelu":
            if kwargs.get("approximate", "none") == "tanh":
                return a0 * 0.5 * (((a0 + a0 * a0 * a0 * 0.044715) * math.sqrt(2.0 / math.pi)).tanh() + 1.0)
            return a0 * 0.5 * ((a0 * (1.0 / math.sqrt(2.0))).erf() + 1.0)
        if name == "tanhshrink":
            return a0 - a0.tanh()
        if name == "softsign":
            return a0 / (a0.abs() + 1.0)
        if name == "mish":
            return a0 * _where(g, a0 > 20.0, a0, a0.minimum(20.0).exp().log1p()).tanh()     # x tanh(softplus(x))
        if name == "hardswish":
            return a0 * (a0 + 3.0).clamp(0.0, 6.0) * (1.0 / 6.0)
        if name == "hardsigmoid":
            return (a0 + 3.0).clamp(0.0, 6.0) * (1.0 / 6.0)
        if name == "logsigmoid":
            return a0.minimum(0.0) - (-a0.abs()).exp().log1p()         # stable on both sides
        if name == "threshold":
            th, val = kwargs.get("threshold", rest[0] if rest else None), kwargs.get("value", rest[1]
                    if len(rest) > 1 else None)
            return _where(g, a0 > float(th), a0, _as_sym(g, float(val)))
        lam = float(kwargs.get("lambd", rest[0] if rest else 0.5))
        if name == "softshrink":
            return _where(g, a0 > lam, a0 - lam, _where(g, a0 < -lam, a0 + lam, _as_sym(g, 0.0)))
        return _where(g, a0.abs() > lam, a0, _as_sym(g, 0.0))                      # hardshrink
    if name in ("softplus",):
        # F.softplus(input, beta=1, threshold=20): linear above the threshold (and no overflow of the exponential there)
        beta = float(kwargs.get("beta", rest[0] if rest else 1.0))
        th = float(kwargs.get("threshold", rest[1] if len(rest) > 1 else 20.0))
        bx = a0 * beta
        return _where(g, bx > th, a0, bx.minimum(th).exp().log1p() / beta)
    if name in ("dropout", "alpha_dropout", "feature_alpha_dropout"):
        if kwargs.get("training", rest[1] if len(rest) > 1 else False):
            raise TraceUnsupported("dropout in training mode")
        return a0
    if name in ("leaky_relu",):
        slope = kwargs.get("negative_slope", rest[0] if rest else 0.01)
        return _where(g, a0 > 0.0, a0, a0 * slope)
    # F.batch_norm(input, running_mean, running_var, weight, bias, training, momentum, eps)
    if name == "batch_norm":
        if kwargs.get("training", rest[4] if len(rest) > 4 else False):
            raise TraceUnsupported("batch_norm in training mode (statistics over the batch)")
        rm, rv = rest[0], rest[1]
        w = kwargs.get("weight", rest[2] if len(rest) > 2 else None)
        b = kwargs.get("bias", rest[3] if len(rest) > 3 else None)
        eps = kwargs.get("eps", rest[6] if len(rest) > 6 else 1e-5)
        if rm is None or rv is None:
            raise TraceUnsupported("batch_norm without running statistics")
        y = (a0 - rm) / (_as_sym(g, rv) + eps).sqrt()
        if w is not None:
            y = y * w
        return y + b if b is not None else y
    if name in ("group_norm", "instance_norm", "embedding", "conv1d", "conv2d"):
        raise TraceUnsupported(f"torch.nn.functional.{name}")
    if name in ("__getitem__",):
        return a0[rest[0]]
    meth = {"absolute": "abs", "negative": "neg", "true_divide": "div", "divide": "div", "multiply": "mul",
            "subtract": "sub",
            "clip": "clamp", "arctan2": "atan2", "linalg_cross": "cross", "linalg_matmul": "matmul",
                    "bitwise_and": "logical_and",
            "bitwise_or": "logical_or", "bitwise_not": "logical_not", "bitwise_xor": "logical_xor"}.get(name, name)
    if a0 is not None and hasattr(SymT, meth) and (not meth.startswith("_") or meth in _DUNDERS):
        f = getattr(a0, meth)
        if callable(f):
            return f(*rest, **kwargs)
        return f
    raise TraceUnsupported(f"torch function {name}")


# ---------------------------------------------------------------------------------------------------------------
# tracing the three callables
# ---------------------------------------------------------------------------------------------------------------
def _flatten_result(r, want, what):
    if isinstance(r, torch.Tensor):
        raise TraceUnsupported(f"{what} returned a constant tensor (it does not depend on its inputs, or left the "
                f"traced ops)")
    if not isinstance(r, SymT):
        raise TraceUnsupported(f"{what} returned {type(r).__name__}")
    a = r.a.reshape(-1)
    if a.size != want:
        raise TraceUnsupported(f"{what} returned {tuple(r.a.shape)} per batch of one, expected {want} value(s)")
    return [int(v) for v in a]


_FACTORIES = {"zeros", "ones", "empty", "full", "tensor", "as_tensor", "eye", "from_numpy", "linspace", "diag",
        "diag_embed",
              "zeros_like", "ones_like", "full_like", "empty_like", "scalar_tensor", "asarray"}
# multiply-adds from which F.linear on a real weight tensor is kept as a layer (below: scalar terms)
DENSE_MIN = 64
_META = {"size", "dim", "numel", "nelement", "stride", "is_floating_point", "is_contiguous", "data_ptr",
        "element_size", "get_device",
         "is_complex", "storage_offset", "__len__", "ndimension", "type", "is_pinned", "__format__", "__repr__",
                 "__str__"}
_RANDOM = {"randn", "rand", "randn_like", "rand_like", "normal", "randint", "bernoulli", "multinomial", "randperm",
        "poisson"}


class _TraceMode(torch.overrides.TorchFunctionMode):
    """While the callables run on symbolic inputs: floating tensors they CREATE (torch.zeros(B, nx) to be filled column
    by column, torch.tensor([...]) constants) become symbolic constants too, so that item assignment of traced values
    into them
    works; random draws are refused (not a function of state, action and timestep); everything else passes through."""
    def __init__(self, g):
        super().__init__()
        self.g = g

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))

        dyn = self.g.dynamic

        def has_sym(v):
            return isinstance(v, (SymT, SymS, _Masked)) or (isinstance(v, (tuple, list))
                    and any(has_sym(e) for e in v)) \
                or (dyn and isinstance(v, torch.Tensor) and id(v) in dyn)

        # a promoted tensor met by a torch function on its own (`self.goal.to(device)`, `goal[None]`)
        def sym_dyn(v):
            if isinstance(v, torch.Tensor) and id(v) in dyn:
                return SymT(self.g, self.g.param_leaves(v))
            if isinstance(v, (tuple, list)) and any(isinstance(e, torch.Tensor) and id(e) in dyn for e in v):
                return type(v)(sym_dyn(e) for e in v)
            return v
        if name in _RANDOM:
            raise TraceUnsupported(f"torch.{name} inside the callable (random draws are not a function of state, "
                    f"action and timestep)")
        if any(has_sym(v) for v in args) or any(has_sym(v) for v in kwargs.values()):
            if any(isinstance(v, _Masked) for v in args):
                raise TraceUnsupported("a masked selection x[mask] passed to a torch function")
            if name == "__setitem__" and isinstance(args[0], torch.Tensor):
                raise TraceUnsupported("item assignment of a traced value into a tensor created outside the traced "
                        "callables")
            args = tuple(SymT(a.g, np.array(a.i)) if isinstance(a, SymS) else a for a in args)
            if dyn:
                if name == "__get__" and isinstance(args[0], torch.Tensor):
                    # a property of a promoted tensor: views become symbolic, metadata (shape, device, dtype, ...) stays
                    # real
                    prop = getattr(getattr(func, "__self__", None), "__name__", "")
                    if prop not in ("T", "mT", "H", "mH", "data", "real"):
                        return func(*args, **kwargs)
                    return getattr(sym_dyn(args[0]), {"H": "T", "mH": "mT", "real": "data"}.get(prop, prop))
                if name in _META and not any(isinstance(a, (SymT, _Masked)) for a in args):
                    return func(*args, **kwargs)
                args = tuple(sym_dyn(a) for a in args)
                kwargs = {k: sym_dyn(v) for k, v in kwargs.items()}
            return _call(self.g, name, args, kwargs)
        out = func(*args, **kwargs)
        srcs = [v for v in list(args) + list(kwargs.values()) if isinstance(v, torch.Tensor)]
        srcs += [e for v in args if isinstance(v, (tuple, list)) for e in v if isinstance(e, torch.Tensor)]
        if srcs:
            for o in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, torch.Tensor):
                    self.g.note_derived(o, srcs)
        if name in _FACTORIES and not srcs and isinstance(out,
                torch.Tensor) and out.is_floating_point() and out.numel() <= 65536 \
                and not out.requires_grad:
            vals = out.detach().cpu().double().numpy()
            return SymT(self.g, np.vectorize(self.g.const, otypes=[np.int64])(vals))
        return out


def trace_callables(dynamics, running_cost, nx, nu, terminal_state_cost=None, step_dependent=False, device=None,
        dtype=None,
                    dynamic=()):
    """-> (Graph, step outputs [nx node ids], cost output id, terminal output id or None)"""
    g = Graph(device=device, dtype=dtype, dynamic=dynamic)

    def xs(kind, n, shape):
        return SymT(g, np.array([g.leaf(kind, i) for i in range(n)], dtype=np.int64).reshape(shape))
    t = SymS(g, g.leaf("t"))
    extra = (t,) if step_dependent else ()
    with torch.no_grad(), _TraceMode(g):
        nxt = dynamics(xs("x", nx, (1, nx)), xs("u", nu, (1, nu)), *extra)
        step_out = _flatten_result(nxt, nx, "dynamics")
        c = running_cost(xs("x", nx, (1, nx)), xs("u", nu, (1, nu)), *extra)
        cost_out = _flatten_result(c, 1, "running_cost")[0]
        term_out = None
        if terminal_state_cost is not None:
            # (1, K=1, T=2, nx): the functor's terminal() sees the LAST state only -- any use of an earlier state or of
            # the actions shows up as a 'y' / 'w' leaf in the result
            st = np.array([[g.leaf("y", i) for i in range(nx)], [g.leaf("x", i) for i in range(nx)]],
                    dtype=np.int64).reshape(1, 1, 2, nx)
            ac = np.array([g.leaf("w", i) for i in range(2 * nu)], dtype=np.int64).reshape(1, 1, 2, nu)
            tr = terminal_state_cost(SymT(g, st), SymT(g, ac))
            term_out = _flatten_result(tr, 1, "terminal_state_cost")[0]
            if _reaches(g, [term_out], ("y", "w")):
                raise TraceUnsupported("terminal_state_cost uses more than the last state")
    return g, step_out, cost_out, term_out


def _reaches(g, roots, kinds):
    seen, stack = set(), list(roots)
    while stack:
        i = stack.pop()
        if i in seen:
            continue
        seen.add(i)
        n = g.nodes[i]
        if n[0] in kinds:
            return True
        if n[0] == "tab":
            stack.append(n[2])
        elif n[0] == "lin":
            stack.extend(g.layers[n[1]]["inputs"])
        elif n[0] not in ("c", "x", "u", "t", "y", "w", "p"):
            stack.extend(n[1:])
    return False


# ---------------------------------------------------------------------------------------------------------------
# code generation
# ---------------------------------------------------------------------------------------------------------------
_FMT1 = {"neg": "(-{0})", "sin": "m_sin({0})", "cos": "m_cos({0})", "tan": "(m_sin({0}) / m_cos({0}))",
        "tanh": "m_tanh({0})",
         "exp": "m_exp({0})", "log": "m_log({0})", "sqrt": "m_sqrt({0})", "abs": "m_abs({0})", "floor": "m_floor({0})",
         "sigmoid": "(T(1) / (T(1) + m_exp(-{0})))", "sign": "(T({0} > T(0)) - T({0} < T(0)))",
         "erf": "m_erf({0})", "atan": "m_atan({0})", "asin": "m_asin({0})", "acos": "m_acos({0})",
                 "sinh": "m_sinh({0})",
         "cosh": "m_cosh({0})", "expm1": "m_expm1({0})", "log1p": "m_log1p({0})", "ceil": "m_ceil({0})",
                 "round": "m_rint({0})",
         "trunc": "m_trunc({0})", "not": "(!{0})"}
_FMT2 = {"add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})",
        "min": "m_min({0}, {1})",
         "max": "m_max({0}, {1})", "pow": "m_pow({0}, {1})", "atan2": "m_atan2({0}, {1})",
         "floormod": "({0} - m_floor({0} / {1}) * {1})", "fmod": "m_fmod({0}, {1})",
         "lt": "({0} < {1})", "le": "({0} <= {1})", "gt": "({0} > {1})", "ge": "({0} >= {1})", "eq": "({0} == {1})",
                 "ne": "({0} != {1})",
         "and": "({0} && {1})", "or": "({0} || {1})", "xor": "({0} != {1})"}


def _lit(v):
    if math.isinf(v):
        return "inf_v<T>()" if v > 0 else "(-inf_v<T>())"
    if math.isnan(v):
        raise TraceUnsupported("NaN constant")
    return f"T({v!r})"


def _deps(g, n):
    """node ids a node's value is computed from"""
    k = n[0]
    if k in ("c", "x", "u", "t", "y", "w", "p"):
        return ()
    if k == "tab":
        return (n[2],)
    if k == "lin":
        return tuple(g.layers[n[1]]["inputs"])
    return n[1:]


def _dense_chains(g, roots):
    """Which dense layers below `roots` feed each other through one elementwise activation and nothing else: -> (tail
    layer -> [(layer, activation format or None) ...] from the chain's head to the tail, set of nodes internal to a
    chain). Layer Lp is fused into L when L's inputs are act(lin(Lp, 0)), act(lin(Lp, 1)), ... in order, with ONE
    activation (a unary function, or max / min / mul / add with a constant) and neither the outputs of Lp nor the
    activations are used
    anywhere else."""
    seen, stack, cons = set(), list(roots), {}
    for r in roots:
        cons[r] = cons.get(r, 0) + 1
    layer_seen = set()
    while stack:
        i = stack.pop()
        if i in seen:
            continue
        seen.add(i)
        n = g.nodes[i]
        # a layer consumes each of its inputs ONCE, however many of its outputs are used
        if n[0] == "lin":
            if n[1] in layer_seen:
                continue
            layer_seen.add(n[1])
        for d in _deps(g, n):
            cons[d] = cons.get(d, 0) + 1
            stack.append(d)
    layers = sorted({g.nodes[i][1] for i in seen if g.nodes[i][0] == "lin"})
    prev = {}
    for L in layers:
        ins = g.layers[L]["inputs"]
        spec, Lp = None, None
        ok = True
        for pos, a in enumerate(ins):
            # walk from the input down to a layer output through operations of ONE operand (unary functions, max / min /
            # mul / add / sub / div with a constant): the activation, innermost operation last in `path`
            path, cur, chain_nodes = [], a, []
            while g.nodes[cur][0] != "lin":
                n = g.nodes[cur]
                # (the distributed form pads a layer's outputs to whole blocks of 16 with zeros, which meet zero weights
                # in the next layer: the activation must be FINITE at 0 -- log(0) or c / 0 would put inf * 0 = NaN into
                # every sum)
                if n[0] in _FMT1 and len(n) == 2 and n[0] not in ("log", "not"):
                    path.append((n[0],))
                    nxt = n[1]
                elif n[0] in ("max", "min", "mul", "add", "sub",
                        "div") and len(n) == 3 and (g.cval(n[1]) is not None) != (g.cval(n[2]) is not None) \
                        and not (n[0] == "div" and g.cval(n[1]) is not None):
                    left_const = g.cval(n[1]) is not None
                    path.append((n[0], g.cval(n[1] if left_const else n[2]), left_const))
                    nxt = n[2] if left_const else n[1]
                else:
                    ok = False
                    break
                chain_nodes.append(cur)
                cur = nxt
                if len(path) > 8:
                    ok = False
                    break
            if not ok:
                break
            sl = g.nodes[cur]
            this = tuple(reversed(path))                        # in the order they are applied to the layer's output
            if sl[2] != pos or (Lp is not None and sl[1] != Lp) or (spec is not None and this != spec):
                ok = False
                break
            if cons.get(cur, 0) != 1 or any(cons.get(c_, 0) != 1 for c_ in chain_nodes):
                ok = False
                break
            spec, Lp = this, sl[1]
        if ok and Lp is not None and g.layers[Lp]["OUT"] % 16 != 0 and not _finite_at_zero(spec):
            # the padded lanes of the wide form would carry NaN / inf into L
            ok = False
        if ok and Lp is not None and g.layers[Lp]["OUT"] == len(ins) and Lp != L:
            prev[L] = (Lp, spec)
    fused_into = {lp: L for L, (lp, _) in prev.items()}
    chains, internal = {}, set()
    for L in layers:
        if L in fused_into:
            continue                                             # not a tail
        chain, cur = [], L
        while True:
            if cur in prev:
                lp, spec = prev[cur]
                chain.append((cur, spec))
                cur = lp
            else:
                chain.append((cur, None))
                break
        # head first; entry i = (layer, activation applied to the PREVIOUS layer's output before this one)
        chain.reverse()
        chains[L] = chain
        for (Lc, _) in chain[:-1]:
            for o in range(g.layers[Lc]["OUT"]):
                internal.add(g.index[("lin", Lc, o)])
        for (Lc, spec) in chain[1:]:
            for a in g.layers[Lc]["inputs"]:
                internal.add(a)
    return chains, internal


def _finite_at_zero(spec):
    """Is the activation path (operations of one operand, in the order they are applied) finite at 0?  The wide
    matrix-core form pads a layer's outputs to whole blocks of 16 with zeros; `sqrt(h - 1)`, `log1p(h - 1)`, `asin(h +
    2)` give NaN / inf on the
    padded lanes, and inf * 0 = NaN in the next layer's products would poison every output of the sample.
    Evaluated on the host in fp64 with numpy's semantics (no exceptions: inf / nan are values)."""
    import numpy as np
    f1 = {"neg": np.negative, "sin": np.sin, "cos": np.cos, "tan": np.tan, "tanh": np.tanh, "exp": np.exp,
            "log": np.log, "sqrt": np.sqrt,
          "abs": np.abs, "floor": np.floor, "sigmoid": lambda v: 1.0 / (1.0 + np.exp(-v)), "sign": np.sign,
                  "atan": np.arctan,
          "asin": np.arcsin, "acos": np.arccos, "sinh": np.sinh, "cosh": np.cosh, "expm1": np.expm1, "log1p": np.log1p,
                  "ceil": np.ceil,
          "round": np.rint, "trunc": np.trunc, "erf": lambda v: np.float64(math.erf(float(v))) if np.isfinite(v) else v}
    f2 = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "min": np.minimum, "max": np.maximum}
    v = np.float64(0.0)
    with np.errstate(all="ignore"):
        for op in spec or ():
            if len(op) == 1:
                if op[0] not in f1:
                    return False
                v = np.float64(f1[op[0]](v))
            else:
                if op[0] not in f2:
                    return False
                c = np.float64(op[1])
                v = np.float64(f2[op[0]](c, v) if op[2] else f2[op[0]](v, c))
    return bool(np.isfinite(v))


def _act_code(spec, var):
    """the activation between two fused layers (operations of one operand, applied in order) on the register array
    `var`, in place"""
    if not spec:
        return ""
    e = f"{var}[i_]"
    for op in spec:
        if len(op) == 1:
            e = _FMT1[op[0]].format(e)
        else:
            e = _FMT2[op[0]].format(_lit(op[1]), e) if op[2] else _FMT2[op[0]].format(e, _lit(op[1]))
    return f"for (int i_ = 0; i_ < (int)(sizeof({var}) / sizeof({var}[0])); ++i_) {var}[i_] = {e}; "


def emit(g, roots, assign=None, ret=False, used=None):
    """C++ statements computing `roots` (node ids): temporaries in topological order, then either `x[i] = ...;`
    assignments (`assign` = list of targets) or `return ...;`.  used: dict collecting the (layer, kind) pairs of the
    dense
    layers the body calls (kind 0: replicated input, 1: distributed input; members of the functor: `layer_members`)."""
    used = {} if used is None else used
    chains, internal = _dense_chains(g, roots) if g.layers else ({}, set())

    def deps(i):
        n = g.nodes[i]
        if n[0] == "lin":
            if n[1] not in chains:
                raise TraceUnsupported("internal: a fused layer's output used outside its chain")
            # a chain's tail depends on the inputs of its head
            return tuple(g.layers[chains[n[1]][0][0]]["inputs"])
        return _deps(g, n)
    order, seen = [], set()
    for r in roots:
        stack = [(r, False)]
        while stack:
            i, done = stack.pop()
            if done:
                order.append(i)
                continue
            if i in seen:
                continue
            seen.add(i)
            stack.append((i, True))
            for a in deps(i):
                stack.append((a, False))
    name = {}
    lines = []
    emitted_chains = set()
    for i in order:
        n = g.nodes[i]
        k = n[0]
        if k == "c":
            name[i] = _lit(n[1])
        elif k == "x":
            name[i] = f"x[{n[1]}]"
        elif k == "u":
            name[i] = f"u[{n[1]}]"
        elif k == "t":
            name[i] = "T(t)"
        elif k == "p":
            name[i] = f"p[{n[1]}]"
        elif k in ("y", "w"):
            raise TraceUnsupported("internal: terminal leaf in a step / cost body")
        elif k == "lin":
            L = n[1]
            if L not in emitted_chains:
                emitted_chains.add(L)
                chain = chains[L]
                head = g.layers[chain[0][0]]
                lines.append(f"T mi{L}[{head['IN']}] = {{{', '.join(name[a] for a in head['inputs'])}}};")
                lines.append(f"T mo{L}[{g.layers[L]['OUT']}];")
                if len(chain) == 1:
                    used[(chain[0][0], 0)] = True
                    lines.append(f"mlp_single(ml{chain[0][0]}_0, mi{L}, mo{L});")
                else:
                    body, prev_var = "{ ", None
                    for ci, (Lc, spec) in enumerate(chain):
                        lay = g.layers[Lc]
                        used[(Lc, 0 if ci == 0 else 1)] = True
                        if ci == 0:
                            body += f"T d{Lc}[mlp_dlen({lay['OUT']}, WX)]; mlp_first(ml{Lc}_0, mi{L}, d{Lc}); "
                        else:
                            body += _act_code(spec, prev_var)
                            if ci < len(chain) - 1:
                                body += f"T d{Lc}[mlp_dlen({lay['OUT']}, WX)]; mlp_mid(ml{Lc}_1, {prev_var}, d{Lc}); "
                            else:
                                body += f"mlp_last(ml{Lc}_1, {prev_var}, mo{L}); "
                        prev_var = f"d{Lc}"
                    lines.append(body + "}")
            name[i] = f"mo{L}[{n[2]}]"
        elif k == "tab":
            vals, N = n[1], len(n[1])
            lines.append(f"const T tab{i}[{N}] = {{{', '.join(_lit(v) for v in vals)}}};")
            lines.append(f"const int ix{i} = (int)({name[n[2]]});")
            lines.append(f"const T v{i} = tab{i}[ix{i} < 0 ? 0 : (ix{i} > {N - 1} ? {N - 1} : ix{i})];")
            name[i] = f"v{i}"
        else:
            ops = [name[a] for a in n[1:]]
            if k in _FMT1:
                e = _FMT1[k].format(*ops)
            elif k == "floormod" and (g.cval(n[2]) or 0.0) > 0.0:
                # a positive constant modulus (angle wrapping): the exact remainder -- k = floor(a / b) from the rounded
                # quotient is off by one at exact multiples of b, where torch's remainder (fmod + sign fix-up) is not
                e = f"m_floormod({ops[0]}, {ops[1]})"
            elif k in _FMT2:
                e = _FMT2[k].format(*ops)
            elif k == "clamp":
                e = f"clampT({ops[0]}, {ops[1]}, {ops[2]})"
            elif k == "select":
                e = f"({ops[0]} ? {ops[1]} : {ops[2]})"
            else:
                raise TraceUnsupported(f"internal: no code for {k}")
            if k in ("lt", "le", "gt", "ge", "eq", "ne", "and", "or", "xor", "not"):
                lines.append(f"const bool v{i} = {e};")
            else:
                lines.append(f"const T v{i} = {e};")
            name[i] = f"v{i}"
    if assign is not None:
        # x[] entries that are read by later assignments are protected by the temporaries above only when every output
        # is a temporary or a leaf other than x[j], j != i: copy leaves first
        outs, copied = [], set()
        for tgt, r in zip(assign, roots):
            if g.nodes[r][0] == "x" and name[r] != tgt:
                if r not in copied:                     # (two outputs may be the same input component: one copy)
                    lines.append(f"const T c{r} = {name[r]};")
                    copied.add(r)
                outs.append((tgt, f"c{r}"))
            else:
                outs.append((tgt, name[r]))
        for tgt, e in outs:
            if tgt != e:
                lines.append(f"{tgt} = {e};")
    if ret:
        lines.append(f"return {name[roots[0]]};")
    return " ".join(lines)


def match_mlp_residual(g, step_roots, cost_root, term_root, nx, nu):
    """Is the traced model the shape the engine's hand-written matrix-core kernel rolls out (csrc/rollout_mlp_split.hip,
    BASELINE configs[3]: x' = x + s (W2 tanh(W1 [x; u] + b1) + b2), cost = sum x^2,
    /root/reference/tests/pendulum_approximate.py:47-67 with
    one hidden layer)?  Structural: exactly one chain of two dense layers with a bare tanh between them, the first
    reading [x_0 .. x_nx-1, u_0 .. u_nu-1] in order, every state component's update `x_i + s * layer2_i` with ONE
    constant s (or none), the cost a diagonal quadratic form sum_i qx_i x_i^2 + sum_n qu_n u_n^2 with constant weights
    (the plain sum x^2 included), no terminal cost.  Returns where W1, b1, W2, b2 sit in the functor's
    parameter vector and s -- or None.  (Whether the kernel exists for (nx, nu, hidden) is the caller's question:
    jit.compile_traced.)"""
    if term_root is not None or len(g.layers) != 2:
        return None
    chains, _ = _dense_chains(g, list(step_roots) + [cost_root])
    if len(chains) != 1:
        return None
    chain = next(iter(chains.values()))
    if len(chain) != 2 or chain[1][1] != (("tanh",),):
        return None
    L1, L2 = chain[0][0], chain[1][0]
    l1, l2 = g.layers[L1], g.layers[L2]
    if l1["IN"] != nx + nu or l2["OUT"] != nx or l2["IN"] != l1["OUT"] or len(step_roots) != nx:
        return None
    if [g.nodes[a] for a in l1["inputs"]] != [("x", i) for i in range(nx)] + [("u", n) for n in range(nu)]:
        return None
    scale = None
    for i, r in enumerate(step_roots):
        n = g.nodes[r]
        if n[0] != "add" or len(n) != 3:
            return None
        if g.nodes[n[1]] == ("x", i):
            other = n[2]
        elif g.nodes[n[2]] == ("x", i):
            other = n[1]
        else:
            return None
        m = g.nodes[other]
        if m == ("lin", L2, i):
            c = 1.0
        elif m[0] == "mul" and len(m) == 3 and g.cval(m[1]) is not None and g.nodes[m[2]] == ("lin", L2, i):
            c = g.cval(m[1])
        elif m[0] == "mul" and len(m) == 3 and g.cval(m[2]) is not None and g.nodes[m[1]] == ("lin", L2, i):
            c = g.cval(m[2])
        else:
            return None
        if scale is not None and c != scale:
            return None
        scale = c
    # cost: a sum of constant multiples of squares of state components and of controls (any association; factors in
    # front of sub-sums distribute): sum_i qx_i x_i^2 + sum_n qu_n u_n^2, nothing else
    qx, qu = [0.0] * nx, [0.0] * nu

    def walk(i, f):
        n = g.nodes[i]
        if n[0] == "add" and len(n) == 3:
            return walk(n[1], f) and walk(n[2], f)
        if n[0] == "neg" and len(n) == 2:
            return walk(n[1], -f)
        if n[0] == "mul" and len(n) == 3:
            a_, b_ = n[1], n[2]
            if a_ == b_ and g.nodes[a_][0] in ("x", "u"):
                (qx if g.nodes[a_][0] == "x" else qu)[g.nodes[a_][1]] += f
                return True
            if g.cval(a_) is not None:
                return walk(b_, f * g.cval(a_))
            if g.cval(b_) is not None:
                return walk(a_, f * g.cval(b_))
            # (c x_i) x_i
            for p_, q_ in ((a_, b_), (b_, a_)):
                m_ = g.nodes[p_]
                if g.nodes[q_][0] in ("x", "u") and m_[0] == "mul" and len(m_) == 3:
                    for c_, v_ in ((m_[1], m_[2]), (m_[2], m_[1])):
                        if g.cval(c_) is not None and v_ == q_:
                            (qx if g.nodes[q_][0] == "x" else qu)[g.nodes[q_][1]] += f * g.cval(c_)
                            return True
        return False
    if not walk(cost_root, 1.0):
        return None
    plain = qx == [1.0] * nx and qu == [0.0] * nu
    return dict(H=int(l1["OUT"]), w1=int(l1["wbase"]), b1=None if l1["bbase"] is None else int(l1["bbase"]),
            w2=int(l2["wbase"]),
                b2=None if l2["bbase"] is None else int(l2["bbase"]), scale=float(scale), **({} if plain
                        else dict(qx=qx, qu=qu)))


def generate(dynamics, running_cost, nx, nu, terminal_state_cost=None, step_dependent=False, device=None, dtype=None,
        dynamic=()):
    """-> dict(step=..., cost=..., terminal=... or None, n_ops=...): the C++ bodies for jit.compile_model.
    device / dtype: what the symbolic inputs report (the controller's; default cpu / float64).
    dynamic: places (watch.Path) whose tensors become run-time parameters instead of constants."""
    g, so, co, to = trace_callables(dynamics, running_cost, nx, nu, terminal_state_cost, step_dependent, device, dtype,
            dynamic)
    used = {}
    step = emit(g, so, assign=[f"x[{i}]" for i in range(nx)], used=used)
    cost = emit(g, [co], ret=True, used=used)
    term = emit(g, [to], ret=True, used=used) if to is not None else None
    members, ctor = layer_members(g, used)
    return dict(step=step, cost=cost, terminal=term, n_ops=len(g.nodes), captured=g.captured,
            param_tensors=g.param_tensors,
                n_params=g.n_params, dynamic=list(dynamic), members=members, ctor=ctor,
                # every numeric constant of the graph (mppi.MPPI._settle_moved)
                numbers=frozenset(n[1] for n in g.nodes if n[0] == "c"),
                mlp_residual=None if step_dependent else match_mlp_residual(g, so, co, to, nx, nu),
                # layers kept as layers
                dense=[dict(IN=g.layers[L]["IN"], OUT=g.layers[L]["OUT"], kind=k) for (L, k) in sorted(used)])


def layer_members(g, used):
    """the functor's dense layers as members (csrc/mlp_wide.hpp MlpLayer) and the constructor statements that bind them
    to their weights in the parameter vector.  In the wide form a layer's A operands and bias live in registers for the
    whole launch (PRE) as long as all layers together stay below ~160 registers per lane; beyond that they are read
    where
    used."""
    regs = 0
    for (L, kind) in used:
        lay = g.layers[L]
        ob = (lay["OUT"] + 15) // 16
        ks = (lay["IN"] + 3) // 4 if kind == 0 else ((lay["IN"] + 15) // 16) * 4
        regs += ob * ks + 4 * ob
    pre = "true" if regs <= 160 else "false"
    members, ctor = [], []
    for (L, kind) in sorted(used):
        lay = g.layers[L]
        members.append(f"MlpLayer<{lay['IN']}, {lay['OUT']}, {kind}, WX, {pre}, T, ParamPtr> ml{L}_{kind};")
        bias = f"p + {lay['bbase']}" if lay["bbase"] is not None else "(ParamPtr)nullptr"
        ctor.append(f"ml{L}_{kind}.load(p + {lay['wbase']}, {bias});")
    return " ".join(members), " ".join(ctor)


def same_functor(a, b):
    """two traces print the same device functor (same bodies: same constants folded in, same parameter reads)"""
    return all(a[k] == b[k] for k in ("step", "cost", "terminal", "n_params"))


def same_param_sources(a, b):
    """... and read their run-time parameters from the same tensors / places"""
    pa, pb = a["param_tensors"], b["param_tensors"]
    if len(pa) != len(pb):
        return False
    for (sa, ba), (sb, bb) in zip(pa, pb):
        if ba != bb or isinstance(sa, PathParam) != isinstance(sb, PathParam):
            return False
        if isinstance(sa, PathParam):
            if sa.path.holder is not sb.path.holder or sa.path.key != sb.path.key or sa.shape != sb.shape:
                return False
        elif sa is not sb:
            return False
    return True


def gather_params(param_tensors, n_params):
    """the model's parameter vector: the trainable tensors' current values, flattened at their bases (on their own
    device)"""
    if not param_tensors:
        return None
    with torch.no_grad():
        ts = [param_tensor(src).detach().reshape(-1).double() for src, _ in param_tensors]
        # (a goal kept on the host beside device weights)
        dev = next((t.device for t in ts if t.device.type != "cpu"), ts[0].device)
        return torch.cat([t.to(dev) for t in ts])


# ---------------------------------------------------------------------------------------------------------------
# verification on the host: the generated bodies compiled by g++ against the callable on random batches
# ---------------------------------------------------------------------------------------------------------------
_HOST = r'''
#include <cmath>
#include <limits>
typedef double T;
template <typename U> static inline U inf_v() { return std::numeric_limits<U>::infinity(); }
static inline T m_sin(T x) { return std::sin(x); }
static inline T m_cos(T x) { return std::cos(x); }
static inline T m_exp(T x) { return std::exp(x); }
static inline T m_tanh(T x) { return std::tanh(x); }
static inline T m_log(T x) { return std::log(x); }
static inline T m_sqrt(T x) { return std::sqrt(x); }
static inline T m_abs(T x) { return std::fabs(x); }
static inline T m_floor(T x) { return std::floor(x); }
static inline T m_min(T a, T b) { return a < b ? a : b; }
static inline T m_max(T a, T b) { return a > b ? a : b; }
static inline T m_pow(T a, T b) { return std::pow(a, b); }
static inline T m_atan2(T a, T b) { return std::atan2(a, b); }
static inline T m_fmod(T a, T b) { return std::fmod(a, b); }
static inline T m_floormod(T a, T b) { T r = std::fmod(a, b); return r < 0 ? r + b : r; }        // b > 0
static inline T m_erf(T x) { return std::erf(x); }
static inline T m_atan(T x) { return std::atan(x); }
static inline T m_asin(T x) { return std::asin(x); }
static inline T m_acos(T x) { return std::acos(x); }
static inline T m_sinh(T x) { return std::sinh(x); }
static inline T m_cosh(T x) { return std::cosh(x); }
static inline T m_expm1(T x) { return std::expm1(x); }
static inline T m_log1p(T x) { return std::log1p(x); }
static inline T m_ceil(T x) { return std::ceil(x); }
static inline T m_rint(T x) { return std::nearbyint(x); }
static inline T m_trunc(T x) { return std::trunc(x); }
static inline T clampT(T x, T lo, T hi) { return std::fmin(std::fmax(x, lo), hi); }
// dense layers kept as layers (csrc/mlp_wide.hpp): on the host the distributed form of a vector is the vector
static const bool WX = false;
typedef const double* ParamPtr;
constexpr int mlp_dlen(int n, bool) { return n; }
template <int IN, int OUT, int KIND, bool WX_, bool PRE, typename U, typename P> struct MlpLayer {
  P w, b;
  void load(P w_, P b_) { w = w_; b = b_; }
  void apply(const U* in, U* out) const {
    for (int o = 0; o < OUT; ++o) { U acc = b ? b[o] : U(0); for (int i = 0; i < IN; ++i) acc += w[o * IN + i] * in[i]; out[o] = acc; }
  }
};
template <class L, int IN, int OUT> static inline void mlp_first(const L& l, const T (&in)[IN], T (&d)[OUT]) { l.apply(in, d); }
template <class L, int IN, int OUT> static inline void mlp_mid(const L& l, const T (&in)[IN], T (&d)[OUT]) { l.apply(in, d); }
template <class L, int IN, int OUT> static inline void mlp_last(const L& l, const T (&in)[IN], T (&d)[OUT]) { l.apply(in, d); }
template <class L, int IN, int OUT> static inline void mlp_single(const L& l, const T (&in)[IN], T (&d)[OUT]) { l.apply(in, d); }
static const int NX = %(nx)d, NU = %(nu)d;
static const double* p;
%(members)s
static inline void step_(T (&x)[NX], const T (&u)[NU], int t) { %(step)s }
static inline T cost_(const T (&x)[NX], const T (&u)[NU], int t) { %(cost)s }
static inline T term_(const T (&x)[NX]) { %(terminal)s }
extern "C" void run(int B, const double* X, const double* U, int t, double* Xn, double* Cc, double* Tc, const double* P) {
  p = P;
  %(ctor)s
  for (int b = 0; b < B; ++b) {
    T x[NX], u[NU];
    for (int i = 0; i < NX; ++i) x[i] = X[b * NX + i];
    for (int n = 0; n < NU; ++n) u[n] = U[b * NU + n];
    Cc[b] = cost_(x, u, t);
    Tc[b] = term_(x);
    step_(x, u, t);
    for (int i = 0; i < NX; ++i) Xn[b * NX + i] = x[i];
  }
}
'''


def evaluate_on_host(code, X, U, nx, nu, t=0):
    """The generated bodies, compiled for the host, on a batch: (next states (B,nx), running costs (B,), terminal costs
    (B,))
    in fp64 -- what the device functor computes, for tests and for looking at a translation by hand."""
    src = _HOST % dict(nx=nx, nu=nu, step=code["step"], cost=code["cost"], terminal=code["terminal"] or "return T(0);",
                       members=code.get("members", ""), ctor=code.get("ctor", ""))
    X, U = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, nx), np.ascontiguousarray(U,
            dtype=np.float64).reshape(-1, nu)
    B = X.shape[0]
    with tempfile.TemporaryDirectory() as d:
        cpp, so = os.path.join(d, "v.cpp"), os.path.join(d, "v.so")
        open(cpp, "w").write(src)
        r = subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", so, cpp], capture_output=True, text=True)
        if r.returncode != 0:
            raise TraceUnsupported("generated code does not compile: " + r.stderr[-400:])
        lib = C.CDLL(so)
        P = gather_params(code.get("param_tensors"), code.get("n_params", 0))
        Pa = np.ascontiguousarray(P.cpu().numpy()) if P is not None else np.zeros(1)
        Xn, Cc, Tc = np.zeros((B, nx)), np.zeros(B), np.zeros(B)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        lib.run(B, p(X), p(U), int(t), p(Xn), p(Cc), p(Tc), p(Pa))
    return Xn, Cc, Tc


def verify_on_host(code, dynamics, running_cost, nx, nu, terminal_state_cost=None, step_dependent=False, B=24,
        rtol=1e-9, horizon=None):
    """Compile the generated bodies for the host and compare with the callables on random batches (fp64).
    Raises TraceUnsupported on any disagreement (the caller keeps the generic path)."""
    src = _HOST % dict(nx=nx, nu=nu, step=code["step"], cost=code["cost"], terminal=code["terminal"] or "return T(0);",
                       members=code.get("members", ""), ctor=code.get("ctor", ""))
    with tempfile.TemporaryDirectory() as d:
        cpp, so = os.path.join(d, "v.cpp"), os.path.join(d, "v.so")
        open(cpp, "w").write(src)
        r = subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", so, cpp], capture_output=True, text=True)
        if r.returncode != 0:
            raise TraceUnsupported("generated code does not compile: " + r.stderr[-400:])
        lib = C.CDLL(so)
        P = gather_params(code.get("param_tensors"), code.get("n_params", 0))
        Pa = np.ascontiguousarray(P.cpu().numpy()) if P is not None else np.zeros(1)
        gen = torch.Generator().manual_seed(12345)
        forms = [("cpu", torch.float64), ("cuda", torch.float64), ("cpu", torch.float32), ("cuda", torch.float32)]
        if not torch.cuda.is_available():
            forms = [f for f in forms if f[0] == "cpu"]
        form = None                                    # (device, dtype) the callables accept: found on the first batch
        # three batches around the origin and one far out (fp64 callables only): rewrites that are only equal where
        # nothing overflows -- log(1 + exp(x)) for softplus -- show up there, matching inf / nan patterns count as
        # agreement
        batches = [(1.0, 0), (3.0, 5), (0.1, 11), (40.0, 2)]
        if step_dependent and horizon is not None:
            # a step-dependent callable is checked at EVERY timestep of the horizon (the last one first: terminal-style
            # terms
            # `c + (t == T - 1) * ...` live there); the host check costs a fraction of a millisecond per batch
            H = int(horizon)
            batches += [(1.0, t) for t in [H - 1] + [t for t in range(H - 1) if t not in (0, 2, 5, 11)][:1023]]
        for scale, t in batches:
            if scale > 10.0 and form is not None and form[1] != torch.float64:
                continue
            if horizon is not None:
                # (a schedule indexed by the timestep is only as long as the horizon)
                t = min(t, int(horizon) - 1)
            X = torch.randn(B, nx, generator=gen, dtype=torch.float64) * scale
            U = torch.randn(B, nu, generator=gen, dtype=torch.float64) * scale
            Xn, Cc, Tc = np.zeros((B, nx)), np.zeros(B), np.zeros(B)
            p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            Xa, Ua = np.ascontiguousarray(X.numpy()), np.ascontiguousarray(U.numpy())
            lib.run(B, p(Xa), p(Ua), int(t), p(Xn), p(Cc), p(Tc), p(Pa))
            extra = (t,) if step_dependent else ()
            with torch.no_grad():
                if form is None:
                    for i, cand in enumerate(forms):   # the callable may have captured device tensors / fp32 weights
                        try:
                            dynamics(X.to(*cand, copy=True), U.to(*cand, copy=True), *extra)
                            form = cand
                            break
                        except RuntimeError:
                            if i == len(forms) - 1:
                                raise
                dev, dt = form
                ref_x = dynamics(X.to(dev, dt, copy=True), U.to(dev, dt, copy=True), *extra).cpu()
                ref_c = running_cost(X.to(dev, dt), U.to(dev, dt), *extra).cpu()
            tol = rtol if dt == torch.float64 else max(rtol, 2e-5)
            if ref_x.numel() != B * nx or ref_c.numel() != B:
                raise TraceUnsupported(f"the callables return {tuple(ref_x.shape)} / {tuple(ref_c.shape)} for a batch "
                        f"of {B}: not one "
                                       f"next state ({nx} values) and one cost per sample")
            pairs = [("dynamics", Xn, ref_x.detach().double().reshape(B, -1).numpy()),
                     ("running_cost", Cc, ref_c.detach().double().reshape(-1).numpy())]
            if terminal_state_cost is not None:
                with torch.no_grad():
                    ref_t = terminal_state_cost(X.to(dev, dt).view(1, B, 1, nx), U.to(dev, dt).view(1, B, 1, nu)).cpu()
                pairs.append(("terminal_state_cost", Tc, ref_t.detach().double().reshape(-1).numpy()))
            for what, got, ref in pairs:
                if got.shape != ref.shape:
                    raise TraceUnsupported(f"{what}: traced result has shape {got.shape}, the callable returns "
                            f"{ref.shape}")
                fin = np.isfinite(ref)
                if not np.array_equal(fin, np.isfinite(got)) or not np.array_equal(np.sign(ref[~fin & ~np.isnan(ref)]),
                        np.sign(got[~fin & ~np.isnan(ref)])) \
                        or not np.array_equal(np.isnan(ref), np.isnan(got)):
                    raise TraceUnsupported(f"{what}: the traced functor and the callable disagree on which results "
                            f"are finite")
                if not fin.any():
                    continue
                s = max(1.0, float(np.abs(ref[fin]).max()))
                err = float(np.abs(got[fin] - ref[fin]).max())
                if not (err <= tol * s):
                    raise TraceUnsupported(f"{what}: traced functor differs from the callable by {err:.3g} (scale "
                            f"{s:.3g})")
    return True


def source_key(code, nx, nu):
    return hashlib.sha256(repr((sorted((k, v) for k, v in code.items() if k in ("step", "cost", "terminal")), nx,
            nu)).encode()).hexdigest()[:12]
