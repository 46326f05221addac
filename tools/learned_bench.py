"""The reference's learned-dynamics network (tests/pendulum_approximate.py:47-53: 3 -> 32 -> 32 -> 2 tanh) through the plain
constructor call at several K: its layers on the matrix cores (sixteen samples per wave, csrc/mlp_wide.hpp) against the
one-lane-per-sample kernels of the same traced functor and the callback loop.   python tools/learned_bench.py [hidden]"""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import jit_fixtures as jf
import pytorch_mppi_amd as pm

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 32
f, q, net = jf.approx_pendulum_callables(hidden=hidden, dtype=torch.float32)
net.cuda()
x0 = torch.tensor([2.5, -0.8]).cuda()


def ms(c, n):
    for _ in range(5):
        c.command(x0)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            c.command(x0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


for K, T in ((1024, 32), (8192, 32), (65536, 32), (8192, 64)):
    mk = lambda auto: pm.MPPI(f, q, 2, torch.tensor(1.0), num_samples=K, horizon=T, device="cuda", lambda_=1.0, u_min=torch.tensor(-2.0),
                              u_max=torch.tensor(2.0), rng="philox", seed=1, auto_jit=auto)
    a = mk(True)
    assert a._model is not None and a._model.wide, a.jit_note
    s = mk(True)
    s._model = copy.copy(a._model); s._model.use_wide = False; s._model.invalidate(); s._problem_cache.clear()
    b = mk(False)
    print(f"hidden {hidden} K {K:6d} T {T}: matrix cores {ms(a, 100):.4f} ms | one lane per sample {ms(s, 20):.4f} ms | callbacks {ms(b, 5):.3f} ms per command",
          flush=True)
