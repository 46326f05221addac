"""Verification of a trace on the host: the generated bodies compiled by g++ and compared with the callables on random batches
(fp64, 1e-9) -- fourth part of the tracer (see pytorch_mppi_amd/trace.py)."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import torch

from .trace_graph import TraceUnsupported
from .trace_emit import gather_params

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
