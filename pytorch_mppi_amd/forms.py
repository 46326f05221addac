"""Which form a command takes and the problem block it hands to the engine (include/mppi_amd.h `MppiProblem`).

`Forms` is the part of `MPPI` that resolves parameters into device vectors (`_vec`), builds and caches the static part
of the problem block (`_static_key`, `_problem`), owns the workspace, moves the state to the device, decides between the
fused path and the callback path (`_needs_generic`), runs the callback path's rollout around the engine's prepare kernel
(/root/reference/src/pytorch_mppi/mppi.py:297-373 is the contract the user's callables see) and materialises the lazily
derived
public arrays."""
import ctypes as C

import torch

from . import _native as N
from ._util import _DT, SpecificActionSampler, _ptr


class Forms:
    """mixin of controller.MPPI (state: `_problem_cache`, `_vec_cache`, `_ws*`, `_generic_memo`; see MPPI.__init__)"""

    def _vec(self, t):
        """(nu,) parameter on device in dtype (0-dim bounds broadcast, mppi.py:124-126).  Cached on
        (tensor identity, in-place version) so that a steady-state command() launches no copy
        kernels for parameters, while assignments / in-place edits by the caller are picked up."""
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        key = (id(t), t._version, self.nu, self.dtype, str(self.d))
        hit = self._vec_cache.get(key)
        if hit is not None and hit[0] is t:
            return hit[1]
        v = t.detach().to(device=self.d, dtype=self.dtype)
        v = v.reshape(-1).expand(self.nu).contiguous() if v.numel() == 1 else v.reshape(-1).contiguous()
        if len(self._vec_cache) > 64:
            self._vec_cache.clear()
        self._vec_cache[key] = (t, v)
        return v

    def _static_key(self, Tn):
        """Identity + in-place version of everything the static part of the problem block is built
        from: a steady-state command() re-uses the cached block and parameter tensors, while
        attribute assignments / in-place edits by the caller (autotune, tests) are picked up."""
        # (this runs on every command: small problems are bound by the host's ~15 us per command, not by the device)
        m = self._model
        a, b, c, d, e, f = self.u_init, self.noise_mu, self._sigma_inv_kernel, self._noise_L, self.u_min, self.u_max
        T_ = torch.Tensor
        return (Tn, self.K_local, self.nx, self.nu, self.k_offset, id(m), m.hidden if m is not None else 0,
                bool(self.noise_abs_cost), bool(self.sample_null_action), int(self.u_per_command),
                float(self.lambda_), float(self.u_scale), int(self.M), float(self.rollout_var_cost),
                float(self.rollout_var_discount), self.seed,
                (id(a), a._version) if isinstance(a, T_) else a, (id(b), b._version) if isinstance(b, T_) else b,
                (id(c), c._version) if isinstance(c, T_) else c, (id(d), d._version) if isinstance(d, T_) else d,
                (id(e), e._version) if isinstance(e, T_) else e, (id(f), f._version) if isinstance(f, T_) else f,
                m._param_version if m is not None else 0)

    def _problem(self, Tn=None, U=None):
        """MppiProblem for this controller: static part cached (see _static_key), U bound fresh."""
        if self.d.type != "cuda":
            raise RuntimeError("pytorch_mppi_amd runs on the MI355X only: construct the controller with "
                               "device='cuda' (there is no CPU compute path)")
        Tn = Tn or self.T
        if getattr(self._model, "_param_tensors", None):
            # a traced model's trainable tensors: re-gathered when written (jit.py)
            self._model.refresh_params()
        key = self._static_key(Tn)
        hit = self._problem_cache.get(Tn)
        if hit is None or hit[0] != key:
            p = N.MppiProblem()
            p.K, p.T, p.nx, p.nu = self.K_local, Tn, self.nx, self.nu
            p.noise_pitch = self._zpitch()
            p.S = 0
            p.dtype = _DT[self.dtype]
            p.k_offset = self.k_offset
            p.model_id = self._model.model_id if self._model is not None else N.MODEL_NONE
            p.hidden = self._model.hidden if self._model is not None else 0
            p.model_flags = self._model.flags() if self._model is not None else 0
            p.sigma_diagonal = int(self._diagonal_sigma)
            p.noise_abs_cost = int(bool(self.noise_abs_cost))
            p.sample_null_action = int(bool(self.sample_null_action))
            p.u_per_command = int(self.u_per_command)
            p.lambda_ = float(self.lambda_)
            p.u_scale = float(self.u_scale)
            p.rollout_samples = int(self.M)
            p.rollout_var_cost = float(self.rollout_var_cost)
            p.rollout_var_discount = float(self.rollout_var_discount)
            p.seed = self.seed
            p.philox_rounds = self.philox_rounds
            keep = dict(
                u_init=self._vec(self.u_init), mu=self._vec(self.noise_mu),
                L=self._noise_L.to(device=self.d, dtype=self.dtype).contiguous(),
                sinv=self._sigma_inv_kernel.to(device=self.d, dtype=self.dtype).contiguous(),
                umin=self._vec(self.u_min), umax=self._vec(self.u_max))
            p.u_init, p.noise_mu = _ptr(keep["u_init"]), _ptr(keep["mu"])
            p.noise_L, p.sigma_inv = _ptr(keep["L"]), _ptr(keep["sinv"])
            p.u_min, p.u_max = _ptr(keep["umin"]), _ptr(keep["umax"])
            if self._model is not None:
                keep["mp"] = self._model.param_blob(self.d, self.dtype)
                p.model_params = _ptr(keep["mp"])
                p.model_params_elems = int(keep["mp"].numel())
                if self._model.process_noise is not None:
                    keep["psd"] = self._model.process_noise.to(device=self.d, dtype=self.dtype).contiguous()
                    p.process_noise_sd = _ptr(keep["psd"])
            hit = (key, p, keep)
            self._problem_cache[Tn] = hit
        # a fresh struct per command (the previous one stays valid for the lazy attributes)
        p = N.MppiProblem.from_buffer_copy(hit[1])
        keep = dict(hit[2])
        Ut = self.U if U is None else U
        if Ut.device != self.d or Ut.dtype != self.dtype or not Ut.is_contiguous():
            Ut = Ut.to(device=self.d, dtype=self.dtype).contiguous()
        keep["U"] = Ut
        p.U = Ut.data_ptr()
        p._keep = keep      # keep the tensors alive as long as the struct
        return p

    def _attach_workspace(self, p):
        key = (p.K, p.T, p.nu, p.num_envs)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = int(N.lib().mppi_workspace_elems(C.byref(p)))
        if self._ws is None or self._ws.numel() < need or self._ws.dtype != self.dtype:
            # zero-filled: the single-launch command keeps its arrival ticket in the last elements
            self._ws = torch.zeros(max(need, 1), device=self.d, dtype=self.dtype)
        p.workspace = self._ws.data_ptr()
        p.workspace_elems = self._ws.numel()

    def _stream(self):
        """The caller's current HIP stream (raw handle; torch.cuda.current_stream() costs ~10 us)."""
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self._dev_index))

    def _host_state_to_device(self, state):
        """A small host-resident state (what a simulator hands back every step) travels inside the launch packet
        of a one-wave kernel (`mppi_upload_small`): ~5 us of host time, no staging buffer to keep alive.  The
        pageable `.to(device)` stalls the host for ~20 us per command -- the whole budget of a small problem."""
        src = state.detach().to(dtype=self.dtype).contiguous()            # host-side cast (a no-op for matching dtypes)
        out = torch.empty(src.shape, dtype=self.dtype, device=self.d)
        N.check(N.lib().mppi_upload_small(src.data_ptr(), src.numel() * src.element_size(), _ptr(out), self._stream()),
                "mppi_upload_small")
        return out

    def _to_state(self, state):
        if not torch.is_tensor(state):
            state = torch.as_tensor(state)
        if state.device.type == "cpu" and self.d.type == "cuda" and 0 < state.numel() * (8
                if self.dtype == torch.float64 else 4) <= 2048:
            state = self._host_state_to_device(state)
        else:
            state = state.to(dtype=self.dtype, device=self.d)             # mppi.py:262-264
        if self.K_local != self.K and tuple(state.shape) == (self.K, self.nx):
            # per-sample initial states (mppi.py:302) of the GLOBAL problem: this shard's rows
            state = state[self.k_offset:self.k_offset + self.K_local]
        return state

    def _sampler_rows(self, p):
        """mppi.py:393-399: rows [null, null+n) come from the sampler; global indices."""
        s = self.specific_action_sampler
        if s is None:
            return
        actions = s.sample_trajectories(self.state, self.info)
        actions = torch.as_tensor(actions).to(device=self.d, dtype=self.dtype).reshape(-1, self.T, self.nu).contiguous()
        i = 1 if self.sample_null_action else 0
        s.register_sample_start_end(i, i + actions.shape[0])
        p.n_sampler_rows = actions.shape[0]
        p.sampler_actions = _ptr(actions)
        p._keep["sampler"] = actions

    def _needs_generic(self):
        # (twice per command, and the host's ~15 us per command are what bounds a small problem: memoised on what it
        # reads)
        m = self._model
        if m is None:
            return True
        s = self.specific_action_sampler
        key = (id(m), m.model_id, self.M, id(s), id(m.process_noise), self.nx, self.nu, self.dtype)
        hit = self._generic_memo
        if hit is not None and hit[0] == key:
            return hit[1]
        v = self._needs_generic_now()
        self._generic_memo = (key, v)
        return v

    def _needs_generic_now(self):
        if self._model is None:
            return True
        if self.M != 1 and not self._fused_multi_ok():
            return True
        if self._model.process_noise is not None and not (self.M > 1 and self._fused_multi_ok()):
            # a model with process noise is stochastic whatever M is (models.NativeModel.with_process_noise); only the
            # fused multi-rollout kernel draws it on the device -- everything else keeps the callables' own noise
            return True
        s = self.specific_action_sampler
        if s is not None and type(s).specific_dynamics is not SpecificActionSampler.specific_dynamics:
            return True      # arbitrary Python post-processing of the dynamics (mppi.py:315-317)
        p_ok = N.model_supported(self._model.model_id, self.nx, self.nu, _DT[self.dtype], self._model.hidden)
        return not p_ok

    def _fused_multi_ok(self):
        """M > 1 rollouts per action sequence inside K1 (csrc/rollout.hpp rollout_stream_multi): MPPI, SMPPI and KMPPI
        (its two-launch form: interpolated raw actions in memory), at most 4 copies of the state per lane; anything else
        runs
        the reference's callback loop."""
        return (1 < self.M <= 4 and self.specific_action_sampler is None and not getattr(self._model, "heavy", False))

    def _fused_state(self, per_sample):
        """Initial state as the fused kernels read it: (K_local,nx) rows or one (nx,) vector.  The
        reference expands anything else to (K, numel) and lets the callbacks cope (mppi.py:305); a
        compiled model has exactly nx state registers, so other sizes are refused."""
        if per_sample:
            return self.state.contiguous()
        if self.state.numel() != self.nx:
            raise ValueError(f"state has shape {tuple(self.state.shape)}; the fused path takes (nx,) = ({self.nx},) "
                             f"or per-sample ({self.K}, {self.nx})")
        return self.state.reshape(-1).contiguous()

    # ------------------------------------------------------------------------------------------
    # generic (callback) path: mppi.py:297-332 around the engine's prepare kernel
    # ------------------------------------------------------------------------------------------
    def _generic_total_cost(self, p, cost_total, st):
        lib = N.lib()
        K, T, nu = self.K_local, self.T, self.nu
        pa = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        noise = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        pert = torch.empty(K, device=self.d, dtype=self.dtype)
        p.perturbed_action, p.noise, p.pert_cost = _ptr(pa), _ptr(noise), _ptr(pert)
        N.check(lib.mppi_prepare(C.byref(p), st), "mppi_prepare")
        p.perturbed_action = p.noise = p.pert_cost = None
        self._perturbed_action, self._noise = pa, noise
        # (no autograd graph through the callbacks: a dynamics network with trainable parameters -- the reference's
        # tests/pendulum_approximate.py -- would otherwise drag requires_grad into cost_total and keep T steps of
        # history)
        with torch.no_grad():
            rollout_cost, self._states, actions = self._callback_rollout(pa)
        self._actions = actions / self.u_scale if actions is not None else None
        torch.add(rollout_cost, pert, out=cost_total)                     # mppi.py:416
        N.check(lib.mppi_cost_block_min(C.byref(p), st), "mppi_cost_block_min")

    def _callback_rollout(self, v):
        """The callback path's K1: the user's callables over the bounded actions `v` (K,T,nu) that `mppi_prepare` wrote,
        one batched call per timestep -> (rollout cost (K,), visited states | None, applied actions | None).

        What the callables see is the reference's contract (mppi.py:297-373), restated around one loop: with M = 1 they
        get (K, .) rows, and the visited states / applied actions are kept -- as (1,K,T,.) -- only when a terminal cost
        wants them; with M > 1 rollouts per action sequence they get M stacked copies of the batch, (M*K, .) rows, every
        copy is kept, and the cost is the mean over the copies plus `rollout_var_cost` x the discounted unbiased
        variance across them.  A sampler's
        `specific_dynamics` post-processes the state after every step (3-D tensors; in the single-copy case it is handed
        the post-dynamics state for both its `next_state` and `state` arguments, in the stacked case the INITIAL states
        for
        `state`)."""
        K, T, nu = v.shape
        M = int(self.M)
        stacked = M > 1
        copies = M if stacked else 1
        start = self.state if tuple(self.state.shape) == (K, self.nx) else self.state.view(1, -1).expand(K, -1)
        if stacked:
            # (M,K,nx): what specific_dynamics gets as `state`
            start = start.repeat(M, 1, 1)
            x = start.reshape(M * K, self.nx)
        else:
            x = start.clone() if start is self.state else start             # (the per-sample states stay the caller's)
        keep = stacked or self.terminal_state_cost is not None
        visited = torch.empty(copies, K, T, self.nx, device=self.d, dtype=self.dtype) if keep else None
        applied = torch.empty(copies, K, T, nu, device=self.d, dtype=self.dtype) if keep else None
        per_copy = torch.zeros(copies, K, device=self.d, dtype=self.dtype)
        spread = torch.zeros(K, device=self.d, dtype=self.dtype) if stacked else None
        sampler = self.specific_action_sampler
        for t in range(T):
            # the callables see scaled actions (mppi.py:313)
            u = self.u_scale * v[:, t]
            if stacked:
                u = u.expand(M, -1, -1)
            rows_u = u.reshape(copies * K, nu) if stacked else u
            x = self._dynamics_fn(x, rows_u, t)
            if sampler is not None:
                if stacked:
                    x = sampler.specific_dynamics(x.reshape(M, K, -1), start.reshape(M, K, -1), u, t).reshape(M * K, -1)
                else:
                    x = sampler.specific_dynamics(x.unsqueeze(0), x.unsqueeze(0), u.unsqueeze(0), t).squeeze(0)
            # the cost of the state just reached ((K,) or (K,1))
            c = self._running_cost_fn(x, rows_u, t).reshape(copies, K)
            per_copy = per_copy + c
            if stacked:
                spread += c.var(dim=0) * self._var_discount_factors[t]
            if keep:
                # dynamics may return more than nx columns (:321)
                visited[:, :, t] = x.reshape(copies, K, -1)[:, :, :self.nx]
                applied[:, :, t] = u
        if stacked:
            per_copy = per_copy + self._terminal_state_cost_fn(visited, applied)
            return per_copy.mean(dim=0) + spread * self.rollout_var_cost, visited, applied
        total = per_copy[0]
        if keep:
            c = self._terminal_state_cost_fn(visited, applied)               # (K,) or (1,K)
            total = total + (c.squeeze(0) if torch.is_tensor(c) and c.dim() > 1 else c)
        return total, visited, applied

    # ------------------------------------------------------------------------------------------
    # lazily materialised public attributes of the fused path (mppi.py:383-385, :411-412)
    # ------------------------------------------------------------------------------------------
    def _materialize(self):
        if self._last is None:
            return
        lib = N.lib()
        p = self._last
        if p.noise_src == N.NOISE_KTN:
            self._convert_noise(p)
        K, T, nu = self.K_local, self.T, self.nu
        pa = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        noise = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        p.perturbed_action, p.noise = _ptr(pa), _ptr(noise)
        N.check(lib.mppi_prepare(C.byref(p), self._stream()), "mppi_prepare")
        p.perturbed_action = p.noise = None
        self._perturbed_action, self._noise = pa, noise

    def _derive_weights(self):
        lam, record = self._lazy_w
        w = torch.exp((-1.0 / lam) * (self.cost_total - record[0]))     # mppi.py:12-13, :256 (beta = record[0])
        self._wnz = w
        self._omega = (1.0 / record[1]) * w                             # :257-258 (eta = record[1])
