"""Per-command time of every controller of the family on C3-sized work (one MI355X):
MPPI / KMPPI / SMPPI with K=65536, T=64, nx=16, nu=12 and MPPI_Batched with N envs x K/N samples,
fused path (Integrator native model), rng = torch-native unless given.
    python tools/variants_bench.py [rng] [name-substring | =exact-name]        (VARIANTS_N: commands per entry, default 50)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm

rng = sys.argv[1] if len(sys.argv) > 1 else "torch-native"
dev = "cuda"
nx, nu, K, T = 16, 12, 65536, 64
m = pm.models.Integrator(nx, nu)
sig = torch.eye(nu)
torch.manual_seed(0)
x0 = torch.randn(nx, device=dev)


only = sys.argv[2] if len(sys.argv) > 2 else ""


N_CMDS = int(os.environ.get("VARIANTS_N", "50"))


HEALTHY = os.environ.get("VARIANTS_HEALTHY_LAMBDA", "1") != "0"


def timeit(name, ctrl, state, n=None):
    n = n or N_CMDS
    if (only[1:] != name) if only.startswith("=") else (only not in name):      # "=MPPI": that entry alone
        return
    if HEALTHY and "peaked" not in name:
        # every entry at ITS OWN healthy lambda = std of its costs (bench.py's recipe): with one fixed lambda the softmax of the
        # unbounded entries collapses (N_eff = 1: K3 skips 98 % of its 64-sample groups) while the bounded ones stay dense, and
        # the table compares K3's sparse path with its dense one instead of the options (profiles/r06_diag_torch_bounds.txt:
        # what round 5 read as "bounds + null action lose a fast form")
        ctrl.command(state)
        ct = ctrl.cost_total
        ctrl.lambda_ = float(ct.float().std(dim=-1).mean()) if ct.dim() > 1 else float(ct.float().std())
    for _ in range(5):
        ctrl.command(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctrl.command(state)
    issue = (time.perf_counter() - t0) / n          # the host's share: how long the Python thread needs to get through a command
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {dt * 1e3:8.4f} ms/command   {K / dt:10.3e} rollouts/s   (host issue {issue * 1e6:6.1f} us/command)", flush=True)


kw = dict(num_samples=K, horizon=T, device=dev, lambda_=50.0)
timeit("MPPI", pm.MPPI(m.dynamics, m.running_cost, nx, sig, rng=rng, **kw), x0)
timeit("MPPI peaked softmax (lambda=0.05)", pm.MPPI(m.dynamics, m.running_cost, nx, sig, rng=rng,
                                                    **{**kw, "lambda_": 0.05}), x0)
timeit("MPPI + u bounds + null action", pm.MPPI(m.dynamics, m.running_cost, nx, sig, rng=rng, u_min=-torch.ones(nu),
                                               u_max=torch.ones(nu), sample_null_action=True, **kw), x0)
mn = pm.models.Integrator(nx, nu).with_process_noise(0.05)
timeit("MPPI M=3 rollouts, process noise", pm.MPPI(mn.dynamics, mn.running_cost, nx, sig, rng=rng, rollout_samples=3,
                                                   rollout_var_cost=0.1, **kw), x0)
full = torch.eye(nu) + 0.1 * torch.ones(nu, nu)
timeit("MPPI full Sigma (Cholesky)", pm.MPPI(m.dynamics, m.running_cost, nx, full, rng=rng, **kw), x0)
timeit("KMPPI S=32", pm.KMPPI(m.dynamics, m.running_cost, nx, sig, num_support_pts=32,
                              kernel=pm.RBFKernel(sigma=2.0), rng=rng, **kw), x0)
timeit("SMPPI", pm.SMPPI(m.dynamics, m.running_cost, nx, sig, rng=rng, action_min=-torch.ones(nu),
                         action_max=torch.ones(nu), w_action_seq_cost=1.0, delta_t=0.1, **kw), x0)
for N in (8, 64):
    b = pm.MPPI_Batched(m.dynamics, m.running_cost, nx, sig, N, num_samples=K // N, horizon=T, device=dev,
                        lambda_=50.0, rng=rng)
    timeit(f"MPPI_Batched N={N} x K={K // N}", b, torch.randn(N, nx, device=dev))
