"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch CPU ops, any dtype) of the reference's
MPPI hot path, written as plain functions over explicit inputs so that the standard-normal
draws ``z`` are an ARGUMENT (no RNG inside).  Every function cites the reference lines it
follows (``mppi.py`` = /root/reference/src/pytorch_mppi/mppi.py).

PARITY PINNING: this restatement is checked (a) bit-for-bit against the LIVE reference in the
build container (tests/test_oracle_reference.py, via oracle/ref_loader.py's z-injection seam)
and (b) against the committed fixtures tests/golden/*.npz that oracle/gen_golden.py produced
by running the live reference -- (b) also runs on machines without /root/reference.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this
module -- as the checker / the timed CPU baseline, never as a compute path of the product.
"""
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch


@dataclass
class Problem:
    """Everything `MPPI.__init__` resolves (mppi.py:45-184), as data."""
    dynamics: Callable
    running_cost: Callable
    nx: int
    noise_sigma: torch.Tensor                 # (nu,nu) or 0-dim
    K: int = 100
    T: int = 15
    lambda_: float = 1.0
    noise_mu: Optional[torch.Tensor] = None
    u_min: Optional[torch.Tensor] = None
    u_max: Optional[torch.Tensor] = None
    u_init: Optional[torch.Tensor] = None
    u_scale: float = 1
    u_per_command: int = 1
    terminal_state_cost: Optional[Callable] = None
    step_dependent_dynamics: bool = False
    sample_null_action: bool = False
    noise_abs_cost: bool = False
    rollout_samples: int = 1                  # M (mppi.py:76, :168)
    rollout_var_cost: float = 0.0             # :77, :170
    rollout_var_discount: float = 0.95        # :78, :171-175
    # resolved in __post_init__
    nu: int = field(init=False)
    dtype: torch.dtype = field(init=False)

    def __post_init__(self):
        s = self.noise_sigma
        self.dtype = s.dtype                                      # mppi.py:88
        self.nu = 1 if s.dim() == 0 else s.shape[0]               # mppi.py:94
        if self.noise_mu is None:
            self.noise_mu = torch.zeros(self.nu, dtype=self.dtype)   # :97-98
        if self.u_init is None:
            self.u_init = torch.zeros_like(self.noise_mu)         # :100-101
        if self.nu == 1:                                          # :104-106
            self.noise_mu = self.noise_mu.view(-1)
            self.noise_sigma = s.view(-1, 1)
        # one-sided bound => symmetric (:112-119); no bound => +-inf so clamp is unconditional (:124-126)
        if self.u_max is not None and self.u_min is None:
            self.u_max = torch.as_tensor(self.u_max)
            self.u_min = -self.u_max
        if self.u_min is not None and self.u_max is None:
            self.u_min = torch.as_tensor(self.u_min)
            self.u_max = -self.u_min
        if self.u_min is None:
            self.u_min = torch.tensor(float("-inf"))
            self.u_max = torch.tensor(float("inf"))
        self.fac = noise_factors(self.noise_sigma)


def noise_factors(sigma):
    """mppi.py:130-139 -- diagonal detection by exact equality; sqrt/inv of the diagonal or
    Cholesky factor + full inverse."""
    diagonal = torch.equal(sigma, torch.diag(torch.diag(sigma)))
    out = {"diagonal": diagonal}
    if diagonal:
        d = torch.diag(sigma)
        out["inv_diag"] = 1.0 / d
        out["sqrt_diag"] = torch.sqrt(d)
        out["sigma_inv"] = torch.diag(out["inv_diag"])
    else:
        out["sigma_inv"] = torch.linalg.inv(sigma)
        out["chol"] = torch.linalg.cholesky(sigma)
    return out


def colour_noise(z, fac, mu):
    """mppi.py:201-206 -- eps = z*sqrt(diag)+mu  |  z @ L^T + mu."""
    if fac["diagonal"]:
        return z * fac["sqrt_diag"] + mu
    return z @ fac["chol"].T + mu


def action_cost(noise, fac, lambda_, abs_cost):
    """mppi.py:186-199 -- lambda * noise * diag^-1  |  lambda * noise @ Sigma^-1 (|noise| if abs)."""
    n = torch.abs(noise) if abs_cost else noise
    if fac["diagonal"]:
        return lambda_ * n * fac["inv_diag"]
    return lambda_ * n @ fac["sigma_inv"]


def shift(U, u_init):
    """mppi.py:232-238"""
    U = torch.roll(U, -1, dims=0)
    U[-1] = u_init
    return U


def overwrite_specific(perturbed, sample_null_action, sampler_actions, T, nu):
    """mppi.py:387-400 -- global row bookkeeping; returns (tensor, (start_idx, end_idx))."""
    i = 0
    if sample_null_action:
        perturbed[i] = 0
        i += 1
    start = end = 0
    if sampler_actions is not None:
        a = sampler_actions.reshape(-1, T, nu)
        perturbed[i:i + a.shape[0]] = a
        start, end = i, i + a.shape[0]
        i += a.shape[0]
    return perturbed, (start, end)


def rollout_costs(p: Problem, state, perturbed_action):
    """mppi.py:297-332 (M == 1 path): cost on the POST-dynamics state with the scaled action."""
    K, T, nu = perturbed_action.shape
    cost = torch.zeros(K, dtype=p.dtype)
    if state.shape == (K, p.nx):
        x = state.clone()                                   # :302-303
    else:
        x = state.view(1, -1).expand(K, -1)                 # :305
    store = p.terminal_state_cost is not None
    if store:
        states = torch.empty(1, K, T, p.nx, dtype=p.dtype)
        actions = torch.empty(1, K, T, nu, dtype=p.dtype)
    for t in range(T):
        u = p.u_scale * perturbed_action[:, t]              # :313
        x = p.dynamics(x, u, t) if p.step_dependent_dynamics else p.dynamics(x, u)   # :314
        c = p.running_cost(x, u, t) if p.step_dependent_dynamics else p.running_cost(x, u)  # :318
        cost = cost + c.reshape(K)                          # :319
        if store:
            states[0, :, t] = x[:, :p.nx]                   # :321
            actions[0, :, t] = u
    if store:
        c = p.terminal_state_cost(states, actions)          # :325
        if torch.is_tensor(c) and c.dim() > 1:
            c = c.squeeze(0)
        cost = cost + c
    else:
        states = actions = None
    return cost, states, actions


def rollout_costs_multi(p: Problem, state, perturbed_action):
    """mppi.py:334-373 (M > 1): every action sequence is rolled out M times through the user's
    (stochastic) callbacks as ONE batch of M*K rows, row = m*K + k; the cost is the mean over the M
    rollouts plus `rollout_var_cost` x the discounted per-step variance over M (unbiased, torch
    `.var(dim=0)`), discount `rollout_var_discount ** t` (:174-175, :364).  `states` / `actions` are
    always stored, (M,K,T,.) (:349-350), and the terminal cost is called unconditionally (:369; the
    default is `lambda states, actions: 0`)."""
    K, T, nu = perturbed_action.shape
    M = p.rollout_samples
    cost_samples = torch.zeros(M, K, dtype=p.dtype)                    # :339-340
    cost_var = torch.zeros(K, dtype=p.dtype)                           # :341
    if state.shape == (K, p.nx):
        s0 = state                                                     # :343-344
    else:
        s0 = state.view(1, -1).expand(K, -1)                           # :346
    s0 = s0.repeat(M, 1, 1)                                            # :348
    states = torch.empty(M, K, T, p.nx, dtype=p.dtype)
    actions = torch.empty(M, K, T, nu, dtype=p.dtype)
    disc = p.rollout_var_discount ** torch.arange(T, dtype=p.dtype)    # :174-175
    x = s0.reshape(M * K, p.nx)
    for t in range(T):
        u = p.u_scale * perturbed_action[:, t].expand(M, -1, -1)       # :354
        uf = u.reshape(M * K, nu)
        x = p.dynamics(x, uf, t) if p.step_dependent_dynamics else p.dynamics(x, uf)          # :356
        c = p.running_cost(x, uf, t) if p.step_dependent_dynamics else p.running_cost(x, uf)  # :361
        c = c.reshape(M, K)
        cost_samples = cost_samples + c                                # :362
        cost_var = cost_var + c.var(dim=0) * disc[t]                   # :363-364
        states[:, :, t] = x.reshape(M, K, -1)[:, :, :p.nx]             # :366
        actions[:, :, t] = u
    if p.terminal_state_cost is not None:
        cost_samples = cost_samples + p.terminal_state_cost(states, actions)   # :369-370
    cost = cost_samples.mean(dim=0) + cost_var * p.rollout_var_cost    # :371-372
    return cost, states, actions


def weights(cost_total, lambda_):
    """mppi.py:254-259 + :12-13 -- beta=min; w=exp(-(1/lambda)(c-beta)); omega = (1/eta) * w."""
    beta = torch.min(cost_total)
    w = torch.exp(-(1 / lambda_) * (cost_total - beta))
    eta = torch.sum(w)
    return (1.0 / eta) * w, w, beta, eta


def compute_rollout_costs(p: Problem, state, perturbed):
    """mppi.py:292-295 `_compute_rollout_costs`: one rollout per action sequence, or M (what MPPI :411, KMPPI :672 and
    SMPPI :564 all call)."""
    if p.rollout_samples > 1:
        return rollout_costs_multi(p, state, perturbed)
    return rollout_costs(p, state, perturbed)


def command(p: Problem, U, state, z, shift_nominal_trajectory=True, sampler_actions=None):
    """One `MPPI.command()` (mppi.py:240-275, :375-417) with injected z of shape (K,T,nu).
    Returns a dict with every public result the reference leaves on `self`."""
    state = torch.as_tensor(state).to(dtype=p.dtype)
    U = U.clone()
    if shift_nominal_trajectory:
        U = shift(U, p.u_init)                              # :249-250
    noise = colour_noise(z, p.fac, p.noise_mu)              # :378
    perturbed = U + noise                                   # :380
    perturbed, slc = overwrite_specific(perturbed, p.sample_null_action, sampler_actions, p.T, p.nu)
    perturbed = torch.clamp(perturbed, p.u_min, p.u_max)    # :383, :419-420
    noise = perturbed - U                                   # :385  (post-clamp noise)
    ac = action_cost(noise, p.fac, p.lambda_, p.noise_abs_cost)   # :409
    rollout_cost, states, actions = compute_rollout_costs(p, state, perturbed)   # :411
    pert_cost = torch.sum(U * ac, dim=(1, 2))               # :415
    cost_total = rollout_cost + pert_cost                   # :416
    omega, w, beta, eta = weights(cost_total, p.lambda_)    # :267
    P = torch.einsum("k,ktn->tn", omega, noise)             # :268
    U_new = U + P                                           # :270
    action = U_new[:p.u_per_command]
    if p.u_per_command == 1:
        action = action[0]                                  # :271-275
    return dict(U=U_new, action=action, cost_total=cost_total, cost_total_non_zero=w, omega=omega,
                noise=noise, perturbed_action=perturbed, beta=beta, eta=eta,
                states=states, actions=(actions / p.u_scale if actions is not None else None),
                sampler_slice=slc, U_shifted=U, rollout_cost=rollout_cost, pert_cost=pert_cost)


# ---------------------------------------------------------------------------------------------
# KMPPI (mppi.py:573-688)
# ---------------------------------------------------------------------------------------------
def rbf_kernel(t, tk, sigma=1.0):
    """mppi.py:587-590 with t:(a,1), tk:(b,1) -> (a,b)"""
    d = torch.sum((t[:, None] - tk) ** 2, dim=-1)
    return torch.exp(-d / (1e-8 + 2 * sigma ** 2))


def kmppi_matrices(T, S, dtype, kernel=rbf_kernel):
    """Constant interpolation operators.  The reference solves K identical systems under vmap
    (mppi.py:630-655); every sample sees the same Tk/Hs/Ktktk (:637-645), so
    W = K(Hs,Tk) @ Ktktk^-1 (T,S) and W_shift = K(Tk+1,Tk) @ Ktktk^-1 (S,S) (:617-619)."""
    Tk = torch.linspace(0, T - 1, int(S), dtype=dtype)      # :637
    Hs = torch.linspace(0, T - 1, int(T), dtype=dtype)      # :639
    Ktktk = kernel(Tk.unsqueeze(-1), Tk.unsqueeze(-1))
    W = torch.linalg.solve(Ktktk, kernel(Hs.unsqueeze(-1), Tk.unsqueeze(-1)), left=False)   # :625
    W_shift = torch.linalg.solve(Ktktk, kernel((Tk + 1).unsqueeze(-1), Tk.unsqueeze(-1)), left=False)
    return W, W_shift, Tk, Hs


def kmppi_command(p: Problem, theta, U, state, z, W, W_shift, shift_nominal_trajectory=True,
                  sampler_actions=None):
    """One `KMPPI.command()` with injected z of shape (K,S,nu) (mppi.py:617-619, :657-688)."""
    state = torch.as_tensor(state).to(dtype=p.dtype)
    U = U.clone()
    if shift_nominal_trajectory:
        U = shift(U, p.u_init)
        theta = W_shift @ theta                             # :619
    noise_S = colour_noise(z, p.fac, p.noise_mu)            # :660
    ctrl_pts = torch.clamp(theta + noise_S, p.u_min, p.u_max)     # :661-663
    noise_theta = ctrl_pts - theta                          # :664
    perturbed = torch.einsum("ts,ksn->ktn", W, ctrl_pts)    # :665 (constant-W form)
    perturbed, slc = overwrite_specific(perturbed, p.sample_null_action, sampler_actions, p.T, p.nu)
    perturbed = torch.clamp(perturbed, p.u_min, p.u_max)    # :668
    noise = perturbed - U                                   # :670
    ac = action_cost(noise, p.fac, p.lambda_, p.noise_abs_cost)
    rollout_cost, states, actions = compute_rollout_costs(p, state, perturbed)   # :672
    cost_total = rollout_cost + torch.sum(U * ac, dim=(1, 2))
    omega, w, beta, eta = weights(cost_total, p.lambda_)
    theta_new = theta + torch.einsum("k,ksn->sn", omega, noise_theta)   # :679-681
    U_new = W @ theta_new                                   # :682
    action = U_new[:p.u_per_command]
    if p.u_per_command == 1:
        action = action[0]
    return dict(U=U_new, theta=theta_new, action=action, cost_total=cost_total, omega=omega,
                cost_total_non_zero=w, noise=noise, noise_theta=noise_theta,
                perturbed_action=perturbed, sampler_slice=slc)


# ---------------------------------------------------------------------------------------------
# SMPPI (mppi.py:451-570): lifted control space
# ---------------------------------------------------------------------------------------------
def smppi_shift(U, A, u_init):
    """mppi.py:488-492"""
    U = torch.roll(U, -1, dims=0)
    U[-1] = u_init
    A = torch.roll(A, -1, dims=0)
    A[-1] = A[-2]
    return U, A


def smppi_command(p: Problem, U, A, state, z, action_min, action_max, w_action_seq_cost=1.0, delta_t=1.0,
                  shift_nominal_trajectory=True, sampler_actions=None):
    """One `SMPPI.command()` with injected z (K,T,nu) (mppi.py:488-492, :523-570).  U is the lifted
    control (action derivative), A the action sequence.  Note the reference quirk kept here: the
    d-action clamp result is stored (`perturbed_control`) but the UNCLAMPED sum feeds
    `perturbed_action` (:536-540)."""
    state = torch.as_tensor(state).to(dtype=p.dtype)
    U, A = U.clone(), A.clone()
    if shift_nominal_trajectory:
        U, A = smppi_shift(U, A, p.u_init)
    noise = colour_noise(z, p.fac, p.noise_mu)              # :533
    perturbed_control = U + noise                           # :535
    perturbed_control_bounded = torch.clamp(perturbed_control, p.u_min, p.u_max)   # :537 (unused below)
    perturbed = A + perturbed_control * delta_t             # :540
    perturbed, slc = overwrite_specific(perturbed, p.sample_null_action, sampler_actions, p.T, p.nu)
    perturbed = torch.clamp(perturbed, action_min, action_max)   # :542
    noise = (perturbed - A) / delta_t - U                   # :544
    ac = action_cost(noise, p.fac, p.lambda_, p.noise_abs_cost)   # :548
    diff = p.u_scale * torch.diff(perturbed, dim=-2)        # :551
    smooth = torch.sum(torch.square(diff), dim=(1, 2)) * w_action_seq_cost   # :552-554
    rollout_cost, states, actions = compute_rollout_costs(p, state, perturbed)   # :556-564
    pert_cost = torch.sum(U * ac, dim=(1, 2))               # :560
    cost_total = rollout_cost + pert_cost + smooth          # :561
    omega, w, beta, eta = weights(cost_total, p.lambda_)
    U_new = U + torch.einsum("k,ktn->tn", omega, noise)     # :511-513
    A_new = A + U_new * delta_t                             # :515
    action = A_new[:p.u_per_command]
    if p.u_per_command == 1:
        action = action[0]
    return dict(U=U_new, action_sequence=A_new, action=action, cost_total=cost_total, omega=omega, noise=noise,
                perturbed_action=perturbed, perturbed_control=perturbed_control_bounded, sampler_slice=slc)
