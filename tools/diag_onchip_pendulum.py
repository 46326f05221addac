"""`onchip/pendulum-nu1-K20000-T48 :: action` sits at 1.44 x of the reference's own fp32 error -- is it the
on-chip command's block-relative beta algebra, or the pendulum's arithmetic (m_sin, the wrapped angle)?  The same case through the
on-chip and the streaming form, each against the fp64 oracle on the same draw, beside the fp32 oracle's own error."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util
import test_gpu_onchip as tg
from oracle import mppi_oracle as orc

kind, nx, nu, K, T = "pendulum", 2, 1, 20000, 48
kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
probe, _, _, _ = tg._make(kind, nx, nu, K, T, False, **kw)
x0 = torch.randn(nx, generator=torch.Generator().manual_seed(3))
probe.command(x0.cuda())
lam = float(probe.cost_total.double().std()) + 1e-3
for seed in (11, 12, 13, 14):
    rows = []
    for onchip in (True, False):
        c, mk, sigma, U0 = tg._make(kind, nx, nu, K, T, onchip, lam=lam, seed=seed, **kw) if "seed" in tg._make.__code__.co_varnames else tg._make(kind, nx, nu, K, T, onchip, lam=lam, **kw)
        Ub = c.U.clone()
        act = c.command(x0.cuda())
        z = gpu_util.device_philox_normals(c, c._call)
        out = []
        for dt in (torch.float64, torch.float32):
            f, q = mk(dt)
            p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma.to(dt), K=K, T=T, lambda_=lam, **{k: v.to(dt) for k, v in kw.items()})
            out.append(orc.command(p, Ub.cpu().to(dt), x0.to(dt), z.to(dt), True))
        r64, r32 = out
        e = lambda name, got: float((got.detach().cpu().double().reshape(-1) - r64[name].double().reshape(-1)).abs().max()) / max(1.0, float(r64[name].abs().max())) if name != "omega" else float((got.detach().cpu().double() - r64[name]).abs().max()) / float(r64[name].abs().max())
        fl = lambda name: float((r32[name].double() - r64[name]).abs().max()) / (max(1.0, float(r64[name].abs().max())) if name != "omega" else float(r64[name].abs().max()))
        rows.append((c.last_draw, {n: (e(n, g), fl(n)) for n, g in (("action", act), ("U", c.U), ("cost_total", c.cost_total), ("omega", c.omega))}))
    for draw, r in rows:
        print(f"seed {seed} {draw:14s} " + "  ".join(f"{n}: err {v[0]:.2e} floor {v[1]:.2e} x{v[0] / max(v[1], 1e-30):.2f}" for n, v in r.items()))
