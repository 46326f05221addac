"""The expression graph a trace records (hash-consed nodes, constants folded), the exceptions of the tracer and the run-time
parameter places -- first part of the tracer (pytorch_mppi_amd/trace.py is the module callers import; it re-exports this)."""
import math
import os

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


# dense layers of at least this many multiply-adds are kept AS LAYERS (Graph "lin" nodes; csrc/mlp_wide.hpp runs them on the matrix cores)
DENSE_MIN = 64


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
