"""Traced callables against the live ones: what keeps a controller honest once plain torch callables have been turned
into a fused device functor (pytorch_mppi_amd/trace.py -> jit.py).

The reference calls the user's `dynamics` / `running_cost` on every command
(/root/reference/src/pytorch_mppi/mppi.py:314, :318): a
functor traced from them has to notice when they would now compute something else.  `JitGlue` is the part of `MPPI` that
traces (`_try_trace`), adopts a background compile (`_adopt_background_model`), watches every place the callables can
read from (`_check_traced` -> watch.StateWatch), re-traces and compares when something moved (`_traced_state_moved`,
`_settle_moved`) and
spot-checks the functor against the callables on the device (`_spot_check`).  Outside SURVEY.md section 8's hot path: an
additive convenience (`auto_jit=`) around the drop-in boundary; a controller built on a `models.NativeModel` never
enters this
file."""
import logging
import os

import torch

from .models import native_model_of

logger = logging.getLogger("pytorch_mppi_amd")


def _auto_jit_mode(v):
    """auto_jit / MPPI_AUTO_JIT -> "sync" | "async" | "0" (anything else is an error: `MPPI_AUTO_JIT=false` must not
    mean on)"""
    if isinstance(v, str):
        v = v.strip().lower()
    m = {True: "sync", "1": "sync", "sync": "sync", "true": "sync", "on": "sync", "yes": "sync", "async": "async",
         False: "0", "": "0", "0": "0", "off": "0", "false": "0", "no": "0", "none": "0"}.get(v)
    if m is None:
        raise ValueError(f"auto_jit / MPPI_AUTO_JIT = {v!r}: expected 'sync', 'async' or '0' (aliases: True/1/on, "
                f"False/0/off/false/no)")
    return m


_TRACE = []


def _trace_module():
    """pytorch_mppi_amd.trace, imported on first use (an `import` statement per command costs a microsecond of a 15 us
    budget)"""
    if not _TRACE:
        from . import trace
        _TRACE.append(trace)
    return _TRACE[0]


class JitGlue:
    """mixin of controller.MPPI (state: `_model`, `_jit_*`, `_traced_user_callables`; see MPPI.__init__)"""

    def _try_trace(self, dynamics, running_cost, terminal_state_cost, step_dependent, background=False, dynamic=(),
                   verify_in_background=False):
        """Plain torch callables -> fused model, or None (generic path; `jit_note` says why).  background: the hipcc run
        happens in a thread unless the object is cached; verify_in_background: so does the host check of the trace (a
        RE-trace in the middle of a control loop must not stall it for the second g++ takes)."""
        import logging
        from . import jit, trace, watch
        log = logging.getLogger("pytorch_mppi_amd")
        try:
            # the places the callables can read from, snapshotted BEFORE they run on symbols: whatever they write there
            # themselves (call counters, `self.last = state`) shows up as a difference and is dropped in _settle_watch
            w = watch.StateWatch([dynamics, running_cost, terminal_state_cost])
            code = jit.trace_and_verify(dynamics, running_cost, self.nx, self.nu, terminal_state_cost, step_dependent,
                    horizon=self.T,
                                        device=self.d, dtype=self.dtype, dynamic=dynamic,
                                                verify=not verify_in_background)
            w.forget([src.path for src, _ in code["param_tensors"] if isinstance(src, trace.PathParam)])
            cached = jit.traced_is_cached(code, self.nx, self.nu, dtype=self.dtype)
            if verify_in_background or (background and not cached):
                # the hipcc run (30 s - 2 min) happens beside the control loop: callbacks until it has finished
                import threading
                box = {"watch": w}

                def work():
                    try:
                        if verify_in_background:
                            jit.verify_traced(code, dynamics, running_cost, self.nx, self.nu, terminal_state_cost,
                                    step_dependent, self.T)
                        box["model"] = jit.compile_traced(code, dynamics, running_cost, self.nx, self.nu,
                                terminal_state_cost,
                                                          step_dependent=step_dependent, dtype=self.dtype)
                    except Exception as e:                      # a failed check / hipcc run: stay on the callbacks
                        box["error"] = e
                th = threading.Thread(target=work, name="pytorch_mppi_amd-jit", daemon=True)
                self._jit_pending = (th, box)
                th.start()
                if not (verify_in_background and cached):
                    self.jit_note = ("generic path for now: the fused kernels of the traced callables are being "
                                     "compiled in the background")
                    log.warning("pytorch_mppi_amd: %s (auto_jit='sync' / MPPI_AUTO_JIT=sync waits for them instead)",
                            self.jit_note)
                return None
            m = jit.compile_traced(code, dynamics, running_cost, self.nx, self.nu, terminal_state_cost,
                    step_dependent=step_dependent,
                                   dtype=self.dtype)
        except trace.TraceUnsupported as e:
            self.jit_note = f"generic path: {e}"
            log.info("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s", e)
            return None
        # a callable that fails on symbolic inputs in its own way, a failed hipcc run, ...
        except Exception as e:
            self.jit_note = f"generic path: {type(e).__name__}: {e}"
            log.info("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s: %s",
                    type(e).__name__, e)
            return None
        m.watch = w
        # (what the callables wrote to their own state while they were traced and checked)
        self._settle_watch(m)
        self._jit_cmds = 0
        self._jit_next_check, self._jit_last_check = 0, None
        self.jit_note = f"fused: traced {m.traced_ops} operations per sample into {m.name}"
        log.info("pytorch_mppi_amd: %s", self.jit_note)
        return m

    # -- traced callables against the live ones
    # ---------------------------------------------------------------------------
    def _callables(self):
        raw = self._traced_user_callables
        # a jit.from_torch model: the user's own callables, not the model's wrappers
        if raw is not None:
            return raw[0], raw[1], raw[2], bool(self.step_dependency)
        return self.F, self.running_cost, self.terminal_state_cost, bool(self.step_dependency)

    def _drop_traced(self, note):
        import logging
        self.jit_note = note
        logging.getLogger("pytorch_mppi_amd").warning("pytorch_mppi_amd: %s", note)
        self._model = None
        self._problem_cache.clear()

    def _check_traced(self, state=None):
        """Once per command, before anything is launched (mppi.py:314,318 call the user's callables on every command: a
        traced functor has to notice when they would now compute something else).  Cheap part, every command: the
        parameter tensors' version counters (`refresh_params`), the version counters of tensors that became constants,
        and the watch over every place the callables can read from (watch.StateWatch.changed, ~0.1 us per place).
        Every `_jit_check_every` commands and on the first command of a newly adopted model: `_spot_check`."""
        m = self._model
        w = getattr(m, "watch", None)
        if w is None:
            return
        trace = _trace_module()
        try:
            if m._param_tensors:
                m.refresh_params()
            moved = w.changed()
        except trace.StaleTrace as e:
            moved, w = None, None
            self._traced_state_moved([], str(e))
            return
        if moved or (m._captured and m.stale()):
            self._traced_state_moved(moved, None)
            return
        n = self._jit_cmds
        self._jit_cmds = n + 1
        if self._jit_check_every > 0 and n >= self._jit_next_check and state is not None \
                and not torch.cuda.is_current_stream_capturing():
            # every `_jit_check_every` commands -- stretched, for problems so small that a check (a millisecond: the
            # user's callables on a batch, a tiny fused rollout, one device sync) would cost more than
            # `_jit_check_share` (1 %) of the
            # time between two of them, to that many commands: a 20 us command is checked every ~6000 commands = 0.12 s
            import time
            t0 = time.perf_counter()
            ok = self._spot_check(state)
            t1 = time.perf_counter()
            last = self._jit_last_check
            every = self._jit_check_every
            if last is not None and n > last[0]:
                period = (t0 - last[1]) / (n - last[0])                 # seconds per command since the previous check
                if period > 0 and self._jit_check_share > 0:
                    every = max(every, min(65536, int((t1 - t0) / (self._jit_check_share * period))))
            self._jit_last_check = (n, t1)
            self._jit_next_check = n + every
            if ok:
                self._jit_unexplained = 0
            else:
                self._traced_state_moved([], "the fused functor and the callables disagree on a random batch")

    def _traced_state_moved(self, moved, why):
        """Something the traced callables can read is not what it was.  Re-trace (symbolic: milliseconds) and compare:
        the same functor source and parameter sources -> irrelevant (forget the places); the same source, parameters
        read from other tensors (a sub-module replaced by one of the same architecture) -> re-bind, no compile; anything
        else
        -> the fused kernels are out of date: back to the callables NOW (the reference's behaviour), new functor
        compiled
        beside the loop with the tensors that moved as run-time parameters."""
        from . import jit, trace
        m, w = self._model, self._model.watch
        dyn, rc, term, sd = self._callables()
        what = why or ("changed: " + w.describe(moved) if moved else "a tensor the traced callables read was modified "
                "in place")
        try:
            code = jit.trace_and_verify(dyn, rc, self.nx, self.nu, term, sd, verify=False, horizon=self.T,
                    device=self.d,
                                        dtype=self.dtype, dynamic=self._jit_dynamic)
        except Exception as e:
            self._drop_traced(f"generic path: the callables' state changed ({what}) and they can no longer be traced: "
                    f"{type(e).__name__}: {e}")
            return
        if trace.same_functor(code, m._code) and why is None:
            self._jit_benign += 1
            if not trace.same_param_sources(code, m._code):
                m.rebind_params(code)
                self._problem_cache.clear()
            self._settle_moved(m, moved, code)
            return
        if why is not None and trace.same_functor(code, m._code) and trace.same_param_sources(code, m._code):
            # a spot-check mismatch that a fresh trace does not explain (a discontinuous cost on a boundary sample,
            # state behind a C extension, a tracer bug): the parameters were re-gathered by the spot-check.  Said aloud,
            # and after three in a row the controller stops trusting the functor: back to the callables, the reference's
            # behaviour (this is the case the spot-check exists for)
            self._jit_benign += 1
            self._jit_unexplained = getattr(self, "_jit_unexplained", 0) + 1
            import logging
            logging.getLogger("pytorch_mppi_amd").warning(
                "pytorch_mppi_amd: %s -- and a fresh trace of the callables prints the same functor (%d in a row)",
                        why, self._jit_unexplained)
            if self._jit_unexplained >= 3:
                self._drop_traced("generic path: the fused functor disagreed with the callables on three spot-checks "
                        "in a row and a fresh "
                                  "trace does not explain it (state behind a C extension? a tracer bug?); "
                                          "ctrl.retrace() tries again")
            else:
                self.jit_note = (f"fused, but a spot-check found an unexplained mismatch ({self._jit_unexplained}): "
                                 f"{m.name}")
            return
        self._jit_retraces += 1
        for path in w.tensors_at(moved or []):
            if not any(q.holder is path.holder and q.key == path.key for q in self._jit_dynamic):
                self._jit_dynamic.append(path)
        self._drop_traced(f"generic path for now: the callables' state changed ({what}); tracing them again")
        if self._jit_retraces > 16:
            self.jit_note = (f"generic path: the callables' state changed {self._jit_retraces} times in ways that "
                    f"change the functor; "
                             f"ctrl.retrace() tries again")
            return
        self._model = self._try_trace(dyn, rc, term, sd, background=True, dynamic=self._jit_dynamic,
                verify_in_background=True)

    def retrace(self, wait=True):
        """Trace the callables again now (what the controller does by itself when it sees their state move); wait=True
        blocks for the host check and the hipcc run unless the kernels are cached.  True when the controller runs
        fused."""
        if self._jit_pending is not None:
            self._jit_pending[0].join()
            self._jit_pending = None
        dyn, rc, term, sd = self._callables()
        if native_model_of(dyn, rc, term) is not None:
            return self._model is not None
        self._model = None
        self._problem_cache.clear()
        self._jit_retraces = 0
        self._model = self._try_trace(dyn, rc, term, sd, background=not wait, dynamic=self._jit_dynamic)
        if self._model is not None:
            self._settle_watch(self._model)
        return self._model is not None

    def _settle_watch(self, m):
        """A traced model is about to serve commands: places that moved since its watch was taken are either the
        callables'
        own doing (they ran on symbols and on the verification batches since) or a real change during the compile."""
        w = m.watch
        moved = w.changed()
        if not moved:
            return True
        from . import jit, trace
        dyn, rc, term, sd = self._callables()
        try:
            code = jit.trace_and_verify(dyn, rc, self.nx, self.nu, term, sd, verify=False, horizon=self.T,
                    device=self.d,
                                        dtype=self.dtype, dynamic=m._code.get("dynamic", ()))
        except Exception:
            return False
        if not trace.same_functor(code, m._code):
            return False
        if not trace.same_param_sources(code, m._code):
            m.rebind_params(code)
        self._settle_moved(m, moved, code, adoption=True)
        return True

    def _settle_moved(self, m, moved, code, adoption=False):
        """A fresh trace prints the same functor although these watched places moved.  Which of them may be forgotten?
        Only those the trace did NOT read (`cost.goal = torch.tensor([2., 1.])` -- same values, a new
        object, what a planner does every cycle -- was judged benign and `GoalCost.goal` dropped from the watch for
        good; the next, real change of the goal then went unseen).  A place keeps being watched, with its present value
        as the new reference, when that value is a tensor / array among the roots of the trace's constants or its
        parameter tensors, or a number / string equal to one of the graph's numeric constants; integers, booleans and
        strings (what Python-level control flow reads
        without leaving a constant behind) are forgotten only after three benign moves in a row, or at adoption (what
        moved while the callables were being traced and verified is their own bookkeeping).  A place re-bound to a NEW
        container or
        object gets the watch rebuilt over the roots, so that what hangs below the new object is watched too."""
        import numpy as np
        from . import trace, watch as watch_mod
        w = m.watch
        read = [c for c, _ in code["captured"]]
        for src, _ in code["param_tensors"]:
            try:
                read.append(trace.param_tensor(src))
            except Exception:
                pass
        numbers = code.get("numbers", frozenset())
        drop, keep, rebuild = [], [], False
        for i in moved:
            path = w.places[i][0]
            v = path.get()
            key = (id(path.holder), path.key if not isinstance(path.holder, watch_mod._Len) else "#len")
            if isinstance(v, (torch.Tensor, np.ndarray)):
                was_read = any(v is r for r in read)
                (keep if was_read else drop).append(i)
                continue
            if isinstance(v, (bool, str, float)) or (isinstance(v, int) and not isinstance(path.holder,
                    watch_mod._Len)):
                # numbers and strings: what Python-level control flow reads leaves no constant behind (`if self.gain >
                # 0.5:`),
                # so "not in the graph" does not mean "not read" -- floats too: kept, with the present
                # value as the new reference, until three benign moves in a row (or adoption)
                n = w.benign.get(key, 0) + 1
                w.benign[key] = n
                in_graph = isinstance(v, (int, bool, float)) and float(v) in numbers
                (drop if (adoption or n >= 3) and not in_graph else keep).append(i)
                continue
            if v is watch_mod._MISSING or v is None or isinstance(path.holder, watch_mod._Len) or isinstance(v,
                    watch_mod._PRIMS):
                drop.append(i)
                continue
            # re-bound to another container / object: the trace may have read what hangs below it
            keep.append(i)
            rebuild = True
        if keep:
            w.resnap(keep)          # (indices stay valid: resnap replaces in place)
        if drop:
            w.drop(drop)
        if rebuild:
            dyn, rc, term, _ = self._callables()
            nw = watch_mod.StateWatch([dyn, rc, term])
            nw.benign = w.benign
            nw.forget(w.dropped_paths)
            nw.dropped, nw.dropped_paths = w.dropped, list(w.dropped_paths)
            m.watch = nw
        m._code = code
        m._captured = list(code["captured"])

    def _spot_check(self, state, samples=64, steps=4):
        """The fused functor against the user's callables on a small random batch ON THE DEVICE (`samples` states around
        the current one, `steps` timesteps of random bounded actions): total costs and visited states of a tiny fused
        rollout against the reference's own loop (mppi.py:297-332) over the same actions.  What the watch cannot see
        ends here: writes through `.data`, state behind C extensions, a tracer bug the host check did not meet.  One
        device
        sync."""
        from .controller import MPPI
        m = self._model
        self._jit_spot_checks += 1
        try:
            m.refresh_params(force=True)              # (a write through .data moves no version counter)
        except Exception:
            return False
        Tp = max(1, min(int(steps), self.T))
        pr = getattr(m, "_probe", None)
        if pr is None or pr.T != Tp:
            term = m.terminal_state_cost if self.terminal_state_cost is not None else None
            pr = MPPI(m.dynamics, m.running_cost, self.nx, self.noise_sigma.reshape(self.nu, self.nu),
                    num_samples=samples, horizon=Tp,
                      device=self.d, terminal_state_cost=term, lambda_=1.0, u_min=self.u_min, u_max=self.u_max,
                              u_scale=self.u_scale,
                      step_dependent_dynamics=bool(self.step_dependency), U_init=torch.zeros(Tp, self.nu,
                              dtype=self.dtype),
                      rng="torch", auto_jit=False)
            pr._want_states = True
            pr._jit_check_every = 0                    # (the probe runs the model under test: it does not check itself)
            m._probe = pr
        if pr._needs_generic() or getattr(m, "_spot_unavailable", False):
            # no fused kernel for this model at the probe's shape: nothing to compare
            return True
        gen = getattr(self, "_spot_gen", None)
        if gen is None:
            gen = self._spot_gen = torch.Generator(device=self.d)
            gen.manual_seed(0x5EED)
        x = self._to_state(state).reshape(-1)
        x = x[:self.nx] if x.numel() >= self.nx else torch.zeros(self.nx, device=self.d, dtype=self.dtype)
        X0 = x + torch.randn(samples, self.nx, device=self.d, dtype=self.dtype, generator=gen) * (0.5 * x.abs() + 1.0)
        z = torch.randn(samples, Tp, self.nu, device=self.d, dtype=self.dtype, generator=gen)
        with torch.no_grad():
            # U = 0: no action cost, cost_total is the rollout's
            pr.U = torch.zeros(Tp, self.nu, device=self.d, dtype=self.dtype)
            pr.inject_noise(z)
            pr.command(X0, shift_nominal_trajectory=False)
            fused_c, fused_x, pa = pr.cost_total, pr.states, pr.perturbed_action

            def reference(dev):
                """the reference's own loop (mppi.py:297-332) over the same actions, its tensors on `dev`"""
                state, ref_c = X0.to(dev).clone(), torch.zeros(samples, device=dev, dtype=self.dtype)
                states = torch.empty(1, samples, Tp, self.nx, device=dev, dtype=self.dtype)
                actions = torch.empty(1, samples, Tp, self.nu, device=dev, dtype=self.dtype)
                pad = pa.to(dev)
                for t in range(Tp):
                    u = self.u_scale * pad[:, t]
                    state = self._dynamics_fn(state, u, t)
                    ref_c = ref_c + self._running_cost_fn(state, u, t).reshape(samples)
                    states[0, :, t] = state[:, :self.nx]
                    actions[0, :, t] = u
                if self.terminal_state_cost is not None:
                    c = self._terminal_state_cost_fn(states, actions)
                    ref_c = ref_c + (c.squeeze(0) if torch.is_tensor(c) and c.dim() > 1 else c)
                return ref_c.to(self.d), states.to(self.d)
            ref = None
            for dev in (self.d, torch.device("cpu")):
                # callables that only work on host tensors (numpy ufuncs on tensors: the reference's own pendulum,
                # tests/pendulum.py:45-46) are checked there; ones that work on neither cannot be checked at all
                try:
                    ref = reference(dev)
                    break
                except Exception:
                    continue
            if ref is None:
                m._spot_unavailable = True
                return True
            ref_c, states = ref
            tol = 2e-3 if self.dtype == torch.float32 else 1e-7
            bad = torch.zeros(samples, dtype=torch.bool, device=self.d)
            for got, ref in ((fused_c, ref_c), (fused_x[0].reshape(samples, -1), states[0].reshape(samples, -1))):
                got, ref = got.reshape(samples, -1), ref.reshape(samples, -1).to(got.dtype)
                fin = torch.isfinite(ref)
                scale = torch.where(fin, ref.abs(), torch.zeros_like(ref)).amax().clamp_min(1.0)
                d = torch.where(fin, (got - ref).abs(), torch.zeros_like(ref))
                bad |= ((d > tol * scale) | (fin != torch.isfinite(got))).any(dim=1)
            # more than a few samples off: not a boundary case of a discontinuous cost
            return int(bad.sum().item()) <= samples // 16

    def _adopt_background_model(self):
        """the background compile of the traced callables (auto_jit="async") has finished: switch to the fused
        kernels"""
        th, box = self._jit_pending
        if th.is_alive():
            return
        self._jit_pending = None
        import logging
        log = logging.getLogger("pytorch_mppi_amd")
        m = box.get("model")
        if m is None:
            self.jit_note = f"generic path: {type(box.get('error')).__name__}: {box.get('error')}"
            log.warning("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s",
                    self.jit_note)
            return
        m.watch = box["watch"]
        if not self._settle_watch(m):
            # the callables' state moved again while these kernels were being compiled: they are already out of date
            self._jit_retraces += 1
            if self._jit_retraces <= 16:
                for path in m.watch.tensors_at(m.watch.changed()):
                    if not any(q.holder is path.holder and q.key == path.key for q in self._jit_dynamic):
                        self._jit_dynamic.append(path)
                dyn, rc, term, sd = self._callables()
                self._model = self._try_trace(dyn, rc, term, sd, background=True, dynamic=self._jit_dynamic,
                        verify_in_background=True)
            return
        self._model = m
        self._jit_cmds = 0
        self._jit_next_check, self._jit_last_check = 0, None
        self._problem_cache.clear()
        self.jit_note = f"fused: traced {m.traced_ops} operations per sample into {m.name} (compiled in the background)"
        log.warning("pytorch_mppi_amd: %s", self.jit_note)

    def wait_for_jit(self, timeout=None):
        """Block until a background compile (auto_jit="async") has finished; True when the controller runs fused
        afterwards."""
        if self._jit_pending is not None:
            self._jit_pending[0].join(timeout)
            self._adopt_background_model()
        return self._model is not None
