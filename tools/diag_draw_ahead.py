"""K3 with the next draw beside it against the plain K3 on the same inputs: which outputs differ, from which command on?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pytorch_mppi_amd as pm

def ctrl(K, T, nx, nu, ahead, **kw):
    model = pm.models.Integrator(nx, nu)
    g = torch.Generator().manual_seed(3)
    c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu) * 0.6, num_samples=K, horizon=T, device="cuda", lambda_=2.0,
                U_init=torch.randn(T, nu, generator=g) * 0.1, **kw)
    c.draw_ahead = ahead
    return c

cases = {"bounds+null": dict(sample_null_action=True, u_min=torch.tensor([-0.4] * 4), u_max=torch.tensor([0.5] * 4)),
         "bounds": dict(u_min=torch.tensor([-0.4] * 4), u_max=torch.tensor([0.5] * 4)),
         "null": dict(sample_null_action=True), "plain": {}}
for name, kw in cases.items():
    for (K, T) in ((20000, 32), (20480, 32), (65536, 32)):
        nx, nu = 8, 4
        x = torch.linspace(-1, 1, nx, device="cuda")
        res = []
        for ahead in (True, False):
            c = ctrl(K, T, nx, nu, ahead, **kw)
            torch.manual_seed(99)
            rec = []
            for i in range(3):
                a = c.command(x).clone()
                rec.append((a, c.cost_total.clone(), c._ws.clone(), c.U.clone(), c._last._keep["z"].clone(), c.last_draw))
            res.append(rec)
        for i in range(3):
            a, b = res[0][i], res[1][i]
            names = ("action", "cost_total", "workspace", "U", "rows")
            d = [n for n, u, v in zip(names, a, b) if not torch.equal(u, v)]
            extra = ""
            if "workspace" in d:
                nz = (a[2] != b[2]).nonzero().flatten()
                extra = f" ws diffs {nz.numel()} first {nz[:6].tolist()} max {float((a[2]-b[2]).abs().max()):.3e}"
            if "rows" in d:
                nz = (a[4] != b[4]).nonzero().flatten()
                extra += f" row diffs {nz.numel()} first {nz[:6].tolist()}"
            print(name, K, T, "cmd", i, a[5], "differ:", d, extra)


# KMPPI (K3 on the theta problem: rows of the support-point draw)
for (K, T, S) in ((24000, 32, 16), (4000, 32, 8)):
    nx, nu = 8, 4
    x = torch.linspace(-1, 1, nx, device="cuda")
    res = []
    for ahead in (True, False):
        model = pm.models.Integrator(nx, nu)
        torch.manual_seed(1)
        c = pm.KMPPI(model.dynamics, model.running_cost, nx, torch.eye(nu) * 0.6, num_samples=K, horizon=T, device="cuda", lambda_=2.0, num_support_pts=S)
        c.draw_ahead = ahead
        torch.manual_seed(99)
        rec = []
        for i in range(3):
            a = c.command(x).clone()
            rec.append((a, c.cost_total.clone(), c._ws.clone(), c.U.clone(), c.theta.clone(), c.last_draw, c._next_hits))
        res.append(rec)
    for i in range(3):
        a, b = res[0][i], res[1][i]
        names = ("action", "cost_total", "workspace", "U", "theta")
        d = [n for n, u, v in zip(names, a, b) if not torch.equal(u, v)]
        print("kmppi", K, T, S, "cmd", i, a[5], b[5], "hits", a[6], "differ:", d, [float((u - v).abs().max()) for n, u, v in zip(names, a, b) if n in d])
