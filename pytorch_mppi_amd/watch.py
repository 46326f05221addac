"""What a traced callable can read besides its arguments -- and whether it still says what it said when it was traced.

The reference evaluates the user's `dynamics` / `running_cost` / `terminal_state_cost` on every command
(/root/reference/src/pytorch_mppi/mppi.py:314, :318, :325), so whatever Python-level state they read is live: an
attribute rebound between two commands (`cost.goal = new_goal`; tests/smooth_mppi.py:54-58 reads `self.goal` on every
call), a Python float gain, a module swapped for another one, a global.  The tracer (trace.py) runs the callables ONCE
and bakes what they read into the device functor.  `StateWatch` closes that gap from the outside: it walks everything
reachable from the callables -- closure cells, defaults, the globals their code names, `__self__`, instance
dictionaries, `nn.Module` parameters / buffers / sub-modules, container items -- and snapshots every place a value can
be read from:

    numbers / strings / None ............ by value
    tensors ............................. by identity, in-place version counter and storage pointer
    small numpy arrays .................. by identity and value
    everything else ..................... by identity (and walked further)

`changed()` re-reads those places (a flat loop, ~0.1 us per place; typical callables have 5 - 60) and returns the ones
that moved.  The controller then re-traces (symbolically: milliseconds): the same functor source means the change was
irrelevant (a call counter, a simulator's own state) and the place is dropped from the list; a different one means the
fused kernels are out of date -- the controller returns to the callables at once and compiles the new functor beside the
loop, with the tensors that changed promoted to RUN-TIME PARAMETERS (`Path`), so that the next `cost.goal = ...` is one
small copy instead of a compile.

Not visible from here: writes through `tensor.data` (no version bump) into an unchanged storage, state behind C
extensions, values the callables compute from the wall clock.  Those are what the controller's periodic spot-check
(mppi.MPPI._spot_check: functor against callables on a small random batch on the device) is for."""
import functools
import types

import numpy as np
import torch

_MISSING = object()
_PRIMS = (int, float, bool, str, bytes, complex, type(None), torch.dtype, torch.device, torch.Size)
# modules whose instances are not user state (walking them finds nothing a callable's result depends on, only noise)
_OPAQUE_MODULE_PREFIXES = ("pytorch_mppi_amd", "threading", "logging", "ctypes", "torch.cuda", "torch._C",
        "torch.distributed",
                           "torch.optim", "torch.utils", "multiprocessing", "concurrent", "socket", "io", "_io",
                                   "matplotlib")
_MODULE_STATE = ("_parameters", "_buffers", "_modules", "training")


class Path:
    """A place a value is read from: `holder[key]` (dictionaries: instance `__dict__`, globals, `Module._parameters`,
    ...;
    lists), `holder.cell_contents` (closure cells) or `getattr(holder, key)`."""
    __slots__ = ("kind", "holder", "key", "_owner")

    def __init__(self, kind, holder, key, owner=None):
        self.kind, self.holder, self.key, self._owner = kind, holder, key, owner

    def get(self):
        try:
            if self.kind == "d":
                return self.holder.get(self.key, _MISSING)
            if self.kind == "l":
                return self.holder[self.key] if self.key < len(self.holder) else _MISSING
            if self.kind == "c":
                return self.holder.cell_contents
            return getattr(self.holder, self.key, _MISSING)
        except ValueError:                      # an empty cell
            return _MISSING

    def __repr__(self):
        h = type(self.holder).__name__
        if self.kind == "d":
            owner = self._owner
            return f"{owner}.{self.key}" if owner else f"{h}[{self.key!r}]"
        if self.kind == "l":
            return f"{h}[{self.key}]"
        if self.kind == "c":
            return f"closure cell '{self.key}'"
        return f"{h}.{self.key}"


def _same_value(a, b):
    return type(a) is type(b) and (a == b or (a != a and b != b))


class StateWatch:
    def __init__(self, roots, max_places=4000, max_depth=10):
        self.places = []            # (Path, snapshot kind, reference object / value)
        self.truncated = False      # the walk stopped at max_places: rely on the spot-check for the rest
        self.dropped = 0
        # the places forgotten by drop(): a watch rebuilt over the same roots forgets them again
        self.dropped_paths = []
        self.benign = {}            # (id(holder), key) -> benign moves in a row (mppi.MPPI._settle_moved)
        self._seen = set()
        self._keep = []             # objects whose id() is in _seen must stay alive for the ids to stay unique
        self._max, self._max_depth = max_places, max_depth
        for r in roots:
            if r is not None:
                self._walk(r, 0, "callable")
        self._compile()

    # -- the walk ------------------------------------------------------------------------------------------------
    def _opaque(self, v):
        if isinstance(v, (types.ModuleType, type, types.BuiltinFunctionType, types.BuiltinMethodType, torch.Generator)):
            return True
        mod = getattr(type(v), "__module__", "") or ""
        return mod.startswith(_OPAQUE_MODULE_PREFIXES)

    @staticmethod
    def _snap(path, v):
        """(path, kind, reference) of a place holding v now"""
        if isinstance(v, _PRIMS) or v is _MISSING:
            return (path, "v", v)
        if isinstance(v, torch.Tensor):
            trainable = isinstance(v, torch.nn.Parameter) or v.requires_grad
            # a trainable tensor's VALUES are run-time parameters of the functor (jit.CustomModel.refresh_params follows
            # its version counter and storage): only its identity is watched here
            return (path, "T" if trainable else "t", (v, v._version, v.data_ptr()))
        if isinstance(v, np.ndarray):
            return (path, "n", (v, v.copy() if v.size <= 4096 else None))
        if isinstance(v, np.generic):
            return (path, "v", v.item())
        return (path, "o", v)

    def _place(self, path, v, depth, label):
        if len(self.places) >= self._max:
            self.truncated = True
            return
        e = self._snap(path, v)
        self.places.append(e)
        if e[1] == "o":
            self._walk(v, depth + 1, label)

    def _dict(self, d, depth, owner, keys=None):
        for k in (list(d.keys()) if keys is None else keys):
            if keys is not None and k not in d:
                continue
            if not isinstance(k, (str, int, float, bool, tuple, type(None))):
                continue
            self._place(Path("d", d, k, owner), d[k], depth, f"{owner}.{k}")
        if keys is None:
            self.places.append((Path("a", _Len(d), "n"), "v", len(d)))

    def _walk(self, v, depth, label):
        if depth > self._max_depth or id(v) in self._seen or self._opaque(v):
            return
        self._seen.add(id(v))
        self._keep.append(v)
        if isinstance(v, types.FunctionType):
            for name, cell in zip(v.__code__.co_freevars, v.__closure__ or ()):
                try:
                    cv = cell.cell_contents
                except ValueError:
                    cv = _MISSING
                self._place(Path("c", cell, name), cv, depth, f"{label}:{name}")
            for attr in ("__defaults__", "__kwdefaults__"):
                dv = getattr(v, attr)
                if dv:
                    self._place(Path("a", v, attr), dv, depth, f"{label}.{attr}")
            names, stack = set(), [v.__code__]
            while stack:                                       # nested lambdas / comprehensions name globals too
                co = stack.pop()
                names.update(co.co_names)
                stack.extend(c for c in co.co_consts if isinstance(c, types.CodeType))
            g = v.__globals__
            self._dict(g, depth, "global", keys=sorted(n for n in names if n in g))
            if v.__dict__:
                self._dict(v.__dict__, depth, label)
        elif isinstance(v, types.MethodType):
            self._place(Path("a", v, "__self__"), v.__self__, depth, f"{label}.__self__")
            self._walk(v.__func__, depth + 1, label)
        elif isinstance(v, functools.partial):
            self._walk(v.func, depth + 1, label)
            self._place(Path("a", v, "args"), v.args, depth, f"{label}.args")
            self._place(Path("a", v, "keywords"), v.keywords, depth, f"{label}.keywords")
        elif isinstance(v, dict):
            self._dict(v, depth, label)
        elif isinstance(v, list):
            for i, e in enumerate(v):
                self._place(Path("l", v, i), e, depth, f"{label}[{i}]")
            self.places.append((Path("a", _Len(v), "n"), "v", len(v)))
        elif isinstance(v, (tuple, frozenset)):
            for i, e in enumerate(v):                          # immutable: the holder's own place watches the identity
                if not isinstance(e, _PRIMS):
                    self._walk(e, depth + 1, f"{label}[{i}]")
        elif isinstance(v, torch.nn.Module):
            d = v.__dict__
            name = type(v).__name__
            self._dict(d, depth, name, keys=[k for k in d if k in _MODULE_STATE or not k.startswith("_")])
            for k in ("_parameters", "_buffers", "_modules"):
                self._walk(d.get(k), depth + 1, f"{name}.{k}")
            self._class_callables(v, depth, name)
        else:
            d = getattr(v, "__dict__", None)
            name = type(v).__name__
            if isinstance(d, dict):
                self._dict(d, depth, name)
            for s in getattr(type(v), "__slots__", ()) or ():
                if isinstance(s, str) and hasattr(v, s):
                    self._place(Path("a", v, s), getattr(v, s), depth, f"{name}.{s}")
            self._class_callables(v, depth, name)

    def _class_callables(self, v, depth, name):
        # the globals named by the methods of a callable object (`__call__`, `forward`) are read live as well
        for meth in ("__call__", "forward"):
            f = getattr(type(v), meth, None)
            if isinstance(f, types.FunctionType):
                self._walk(f, depth + 1, f"{name}.{meth}")

    # -- the check -----------------------------------------------------------------------------------------------
    def _compile(self):
        """split the places by what has to be compared and how they are read (tight loops in `changed`: most places are
        dictionary slots compared by identity, ~70 ns each)"""
        self._ident_d, self._ident_o, self._vals_d, self._vals_o = [], [], [], []
        self._tens, self._arrs, self._lens = [], [], []
        for i, (path, kind, ref) in enumerate(self.places):
            if isinstance(path.holder, _Len):
                self._lens.append((i, path.holder.c, ref))
                continue
            if kind == "v":
                # numbers compare by value AND type (True == 1 == 1.0); other immutables by value
                if path.kind == "d":
                    self._vals_d.append((i, path.holder, path.key, ref, type(ref)))
                else:
                    self._vals_o.append((i, path, ref))
                continue
            obj = ref[0] if kind in ("t", "T", "n") else ref
            if path.kind == "d":
                self._ident_d.append((i, path.holder, path.key, obj))
            else:
                self._ident_o.append((i, path, obj))
            if kind == "t":
                self._tens.append((i, ref[0], ref[1], ref[2]))
            elif kind == "n" and ref[1] is not None:
                self._arrs.append((i, ref[0], ref[1]))

    def changed(self):
        """indices of the places whose value is not what it was at snapshot time ([] = nothing moved)"""
        out = []
        M = _MISSING
        for i, d, k, ref in self._ident_d:
            if d.get(k, M) is not ref:
                out.append(i)
        for i, d, k, ref, ty in self._vals_d:
            v = d.get(k, M)
            if v is not ref and (type(v) is not ty or not (v == ref or (v != v and ref != ref))):
                out.append(i)
        for i, c, n in self._lens:
            if len(c) != n:
                out.append(i)
        for i, path, ref in self._ident_o:
            if path.get() is not ref:
                out.append(i)
        for i, path, ref in self._vals_o:
            if not _same_value(path.get(), ref):
                out.append(i)
        for i, t, ver, ptr in self._tens:
            if t._version != ver or t.data_ptr() != ptr:
                out.append(i)
        for i, a, copy in self._arrs:
            if not np.array_equal(a, copy):
                out.append(i)
        return out

    def describe(self, idx, limit=4):
        s = ", ".join(repr(self.places[i][0]) for i in idx[:limit])
        return s + (f", ... ({len(idx)} places)" if len(idx) > limit else "")

    def drop(self, idx):
        """forget these places (they change without changing what the callables compute)"""
        gone = set(idx)
        self.dropped += len(gone)
        self.dropped_paths.extend(self.places[i][0] for i in gone if i < len(self.places))
        self.places = [p for i, p in enumerate(self.places) if i not in gone]
        self._compile()

    def forget(self, paths):
        """drop the places at these paths (tensors promoted to run-time parameters: jit.CustomModel.refresh_params
        follows
        what sits there from now on -- identity, version counter, storage -- and re-gathers instead of re-tracing)"""
        idx = [i for i, (p, _, _) in enumerate(self.places)
               if any(p.kind == q.kind and p.holder is q.holder and p.key == q.key for q in paths)]
        if idx:
            n = self.dropped
            self.drop(idx)
            self.dropped = n

    def resnap(self, idx):
        """take the current value of these places as the new reference (what hangs below a new object is not walked)"""
        for i in idx:
            path = self.places[i][0]
            self.places[i] = self._snap(path, path.get())
        self._compile()

    def tensors_at(self, idx, max_numel=4096):
        """the changed places that hold (non-trainable) floating tensors NOW: candidates for run-time parameters"""
        out = []
        for i in idx:
            path, kind, ref = self.places[i]
            v = path.get()
            if kind in ("t", "o", "v") and isinstance(v,
                    torch.Tensor) and v.is_floating_point() and 0 < v.numel() <= max_numel \
                    and not (isinstance(v, torch.nn.Parameter) or v.requires_grad) and path.kind in ("d", "c", "l"):
                out.append(path)
        return out


class _Len:
    """length of a container as an attribute (so that it fits the Path protocol)"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = c

    @property
    def n(self):
        return len(self.c)
