"""Dev helper (GPU): soak of the two-wave on-chip K1 (csrc/rollout_onchip_pair.hpp) -- its hand-over between the waves of a pair rests on
LDS release / acquire and per-pair progress counters, no workgroup barrier.  Two controllers on the same stream of commands, one on the
two-wave kernel, one on the one-wave kernel (MPPI_ONCHIP_PAIR flipped in the environment before every command: the engine reads it at
every launch); every output compared bit for bit every 200 commands; a moving state, healthy and peaked softmax, MPPI and SMPPI, and a
second pair of controllers on another HIP stream running at the same time."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, pytorch_mppi_amd as pm
from pytorch_mppi_amd import _native as N
N_CMD = int(os.environ.get("SOAK_N", "40000"))
m = pm.models.Integrator(16, 12)
lib = N.lib()


def mk(cls, K, T, lam, seed, **kw):
    # (U_init given: without it the nominal sequence starts as a draw from torch's global generator, mppi.py:161 -- two controllers, two draws)
    c = cls(m.dynamics, m.running_cost, 16, torch.eye(12) * 0.5, num_samples=K, horizon=T, device="cuda", lambda_=lam, rng="philox", seed=seed,
            U_init=torch.zeros(T, 12), **kw)
    c.philox_onchip = True
    return c


def soak(name, cls, K, T, lam, n, **kw):
    a, b = mk(cls, K, T, lam, 5, **kw), mk(cls, K, T, lam, 5, **kw)
    x = torch.randn(16, generator=torch.Generator().manual_seed(1)).cuda() * 0.3
    t0, bad, checks = time.time(), 0, 0
    p0 = int(lib.mppi_stat_onchip_pair_launches())
    for i in range(n):
        os.environ["MPPI_ONCHIP_PAIR"] = "1"
        ua = a.command(x)
        os.environ["MPPI_ONCHIP_PAIR"] = "0"
        ub = b.command(x)
        if i % 200 == 0:
            ok = torch.equal(ua, ub) and torch.equal(a.U, b.U) and torch.equal(a.cost_total, b.cost_total)
            bad += 0 if ok else 1
            checks += 1
            x = x + 0.01 * ua.mean()            # a moving state: every block of commands differs
    torch.cuda.synchronize()
    print(f"{name}: {n} commands x 2 controllers in {time.time() - t0:.1f} s: {bad} mismatching checks of {checks}; "
          f"two-wave launches {int(lib.mppi_stat_onchip_pair_launches()) - p0}; forms {a.last_draw} / {b.last_draw}", flush=True)
    return bad


bad = soak("MPPI C3 healthy", pm.MPPI, 65536, 64, 3000.0, N_CMD)
bad += soak("MPPI C3 peaked", pm.MPPI, 65536, 64, 0.5, N_CMD // 4)
bad += soak("MPPI K=50000 T=48 bounds null row", pm.MPPI, 50000, 48, 2000.0, N_CMD // 4, sample_null_action=True,
            u_min=torch.tensor([-0.4] * 12), u_max=torch.tensor([0.6] * 12))
bad += soak("SMPPI C3", pm.SMPPI, 65536, 64, 3000.0, N_CMD // 4, w_action_seq_cost=2.0, delta_t=0.5)
# two streams at once: a second soak on its own stream from a second thread (the kernels cannot share a CU -- 150 KB of LDS each -- but
# their workgroups interleave on the chip)
res = []


def other():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a = mk(pm.MPPI, 49152, 64, 3000.0, 7)
        x = torch.zeros(16).cuda()
        r0 = None
        for i in range(N_CMD // 8):
            u = a.command(x)
        s.synchronize()
        res.append(a.U.clone())


os.environ["MPPI_ONCHIP_PAIR"] = "1"
t = threading.Thread(target=other)
t.start()
c = mk(pm.MPPI, 65536, 64, 3000.0, 9)
x = torch.zeros(16).cuda()
for i in range(N_CMD // 8):
    c.command(x)
t.join()
torch.cuda.synchronize()
# the same two sequences alone
d = mk(pm.MPPI, 49152, 64, 3000.0, 7)
for i in range(N_CMD // 8):
    d.command(x)
e = mk(pm.MPPI, 65536, 64, 3000.0, 9)
for i in range(N_CMD // 8):
    e.command(x)
torch.cuda.synchronize()
ok = torch.equal(res[0], d.U) and torch.equal(c.U, e.U)
print(f"two controllers on two streams / threads at once, {N_CMD // 8} commands each: {'the same U as alone' if ok else 'MISMATCH'}")
print("SOAK", "OK" if bad == 0 and ok else "FAILED")
