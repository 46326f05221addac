"""Why does rng="torch" + u bounds + null action cost more than plain MPPI at C3 (0.1342 vs 0.1153 ms)?
Separates the two options, with and without the next draw inside K3 (MPPI_DRAW_AHEAD), and reports what the softmax looks like
(samples whose weight is exactly zero, 64-sample groups K3 may skip) -- the draw-ahead K3 runs the same k3_diag_block either way.
    python tools/diag_torch_bounds.py [out.txt]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def child(which, lam):
    import torch
    import pytorch_mppi_amd as pm
    nx, nu, K, T = 16, 12, 65536, 64
    m = pm.models.Integrator(nx, nu)
    torch.manual_seed(0)
    x0 = torch.randn(nx, device="cuda")
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=lam, rng="torch")
    if "b" in which:
        kw.update(u_min=-torch.ones(nu), u_max=torch.ones(nu))
    if "n" in which:
        kw.update(sample_null_action=True)
    c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), **kw)
    for _ in range(8):
        c.command(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        c.command(x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    w = c.cost_total_non_zero
    zero = float((w == 0).float().mean())
    live = float((w.reshape(-1, 64) != 0).any(dim=1).float().mean())
    n_eff = 1.0 / float((c.omega.double() ** 2).sum())
    print(f"RESULT {dt * 1e3:.4f} {zero:.4f} {live:.4f} {n_eff:.1f} {c.last_draw}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], float(sys.argv[3]))
        sys.exit(0)
    lines = ["# tools/diag_torch_bounds.py: C3 shape, rng=torch, per command (100 commands back to back)",
             f"# {'options':<22} {'lambda':>7} {'draw-ahead':>10} {'ms/cmd':>8} {'w == 0':>8} {'live groups':>11} {'N_eff':>9}  draw"]
    for lam in (50.0, 5000.0):
        for which, name in (("", "plain"), ("b", "u bounds"), ("n", "null action"), ("bn", "bounds + null action")):
            for ahead in ("1", "0"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which or "-", str(lam)],
                                   env=dict(os.environ, MPPI_DRAW_AHEAD=ahead), capture_output=True, text=True, timeout=300)
                res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
                if not res:
                    lines.append(f"  {name:<22} {lam:7.0f} {ahead:>10} FAILED {r.stderr[-300:]}")
                    continue
                _, ms, zero, live, neff, draw = res[0].split()
                lines.append(f"  {name:<22} {lam:7.0f} {ahead:>10} {float(ms):8.4f} {float(zero):8.4f} {float(live):11.4f} {float(neff):9.1f}  {draw}")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)
