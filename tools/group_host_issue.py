"""Host time of ONE command of a device group (`MPPI(..., devices=[...])`, pytorch_mppi_amd/group.py) -- VERDICT r05 next #1.

    python tools/group_host_issue.py [out.txt]

(a) host-only issue: N shard objects on a problem so small (K = 256 per shard, T = 8) that the GPU's share is a few microseconds --
    what is left is the host: Python's N problem blocks + the hand-over + the launches.  N = 2, 4, 8, with the engine's worker
    threads (csrc/group.hip, one per device) and, for comparison, the one-thread form of round 5 (MPPI_GROUP_THREADS=0: every
    shard's launches from the Python thread).  The box has ONE GPU: every shard is listed on device 0 (the "staged" exchange) --
    the host path is the product's, the exchange is copies instead of RCCL.
(b) the rig at C3: devices=[0] * N with K = N x 65536 (N = 2, 8) against the unsharded controller at K = 65536 on the same device: N
    shards back to back on one GPU cannot beat N x the single command; how close the group comes says whether the host is in the
    way -- and issue_us (bursts of 20 commands from an idle queue) is the host's share of a C3-sized sharded command.
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one(n, K_per, T, nx, nu, steps, threads):
    import torch
    import pytorch_mppi_amd as pm
    m = pm.models.Integrator(nx, nu)
    torch.manual_seed(0)
    U0 = torch.randn(T, nu) * 0.02
    devs = [0] * n if n > 1 else None
    c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K_per * n, horizon=T, device="cuda", lambda_=20.0,
                U_init=U0, rng="philox", seed=7, devices=devs)
    x = torch.randn(nx, device="cuda")
    for _ in range(50):
        c.command(x)
    torch.cuda.synchronize()
    best, best_issue, best_wait = 1e9, 1e9, 0.0
    for _ in range(7):
        torch.cuda.synchronize()
        w0 = getattr(c, "wait_seconds", 0.0)
        t0 = time.perf_counter()
        for _ in range(steps):
            c.command(x)
        t_issue = time.perf_counter() - t0
        w1 = getattr(c, "wait_seconds", 0.0)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        best = min(best, t_all / steps)
        if t_issue / steps < best_issue:
            best_issue, best_wait = t_issue / steps, (w1 - w0) / steps
    # (`steps` is small enough for every packet to fit the queue: the host never waits for the GPU while it issues, so
    #  issue_us is the host's own time per command even where the GPU -- one device for all shards here -- takes longer)
    return best_issue * 1e6, best * 1e6, best_wait * 1e6, getattr(c, "issue", "single controller")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        n, K_per, T, nx, nu, steps = (int(v) for v in sys.argv[2:8])
        iss, per, wait, how = one(n, K_per, T, nx, nu, steps, None)
        print(f"RESULT {iss:.2f} {per:.2f} {wait:.2f} {how}")
        return
    out = sys.argv[1] if len(sys.argv) > 1 else None
    lines = ["# tools/group_host_issue.py -- host time per command of a device group (one MI355X; every shard on device 0, staged exchange)",
             "# issue_us = time for the Python thread to get through a command (no synchronisation; (a): 100 commands per burst, so that "
             "the queue never fills and the host never waits for the GPU); cmd_us = commands back to back, synchronised at the end (best of 7)"]

    def run(n, K_per, T, nx, nu, steps, threads):
        env = dict(os.environ, MPPI_GROUP_THREADS="1" if threads else "0")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(K_per), str(T), str(nx), str(nu), str(steps)],
                           env=env, capture_output=True, text=True, timeout=600)
        for ln in r.stdout.splitlines():
            if ln.startswith("RESULT"):
                _, iss, per, wait, how = ln.split(" ", 4)
                return float(iss), float(per), float(wait), how
        raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])

    lines.append("# (a) host-only issue: K = 256 per shard, T = 8, nx = 8, nu = 4")
    lines.append("# caller_us = issue_us - the time inside mppi_group_wait; workers_us = that wait: the workers issuing their launches -- here, with "
                 "every shard on ONE device, one behind the other on that device's queue; N distinct devices issue in parallel (~ workers_us / N)")
    lines.append(f"# {'shards':>6} {'form':<46} {'issue_us':>9} {'caller_us':>9} {'workers_us':>10} {'cmd_us':>8} {'caller + workers/N':>19}")
    iss1, per1, _, _ = run(1, 256, 8, 8, 4, 100, True)
    lines.append(f"  {1:6d} {'single controller (no group)':<46} {iss1:9.2f} {'':>9} {'':>10} {per1:8.2f}")
    for n in (2, 4, 8):
        for threads in (True, False):
            iss, per, wait, how = run(n, 256, 8, 8, 4, 100, threads)
            lines.append(f"  {n:6d} {('engine worker threads' if threads else 'one Python thread (r05 form)'):<46} {iss:9.2f} {iss - wait:9.2f} {wait:10.2f} "
                         f"{per:8.2f} {(iss - wait + wait / n) if threads else iss:19.2f}")
    lines.append("# (b) the rig at C3 (nx = 16, nu = 12, T = 64): K = 65536 per shard")
    iss1, per1, _, _ = run(1, 65536, 64, 16, 12, 20, True)
    lines.append(f"  {1:6d} {'single controller (no group)':<46} {iss1:9.2f} {'':>9} {'':>10} {per1:8.2f}")
    for n in (2, 8):
        for threads in (True, False):
            iss, per, wait, how = run(n, 65536, 64, 16, 12, 20, threads)
            lines.append(f"  {n:6d} {('engine worker threads' if threads else 'one Python thread (r05 form)'):<46} {iss:9.2f} {iss - wait:9.2f} {wait:10.2f} "
                         f"{per:8.2f}   = {per / per1:.2f} x the single command ({n}.00 = GPU-bound on one device)")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if out:
        open(out, "w").write(txt)


if __name__ == "__main__":
    main()
