"""Tracing the three callables and printing the recorded arithmetic as the C++ `step` / `cost` / `terminal` bodies that
jit.compile_model compiles around csrc/rollout.hpp; the matcher that recognises a 2-layer residual MLP (-> the built-in matrix-core
kernels) -- third part of the tracer (see pytorch_mppi_amd/trace.py)."""
import math

import numpy as np
import torch

from .trace_graph import Graph, PathParam, TraceUnsupported, param_tensor
from .trace_sym import SymS, SymT, _Masked, _call

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
