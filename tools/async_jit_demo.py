"""auto_jit="async" on the GPU: a controller built on callables the engine has never seen returns at once, runs the callback
loop while hipcc works in a background thread, and adopts the fused kernels when they are ready.
    python tools/async_jit_demo.py > gpurun_out/async_jit_demo.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_mppi_amd as pm

c0 = 0.1 + (time.time() % 1000) * 1e-7          # a constant nobody has compiled before: a fresh object every run
f = lambda s, a: s + c0 * torch.tanh(a) + 0.01 * torch.cos(s * 1.5)
q = lambda s, a: (s ** 2).sum(-1) + 0.3 * (a ** 2).sum(-1)
t0 = time.perf_counter()
c = pm.MPPI(f, q, 3, torch.eye(3), num_samples=8192, horizon=32, device="cuda", auto_jit="async")
print("constructor returned after %.2f s: %s" % (time.perf_counter() - t0, c.jit_note), flush=True)
x = torch.zeros(3, device="cuda")
n_cb, t_cb, adopted_at = 0, 0.0, None
while time.perf_counter() - t0 < 240:
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c.command(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    if c._model is not None:
        adopted_at = time.perf_counter() - t0
        break
    n_cb, t_cb = n_cb + 1, t_cb + dt
print("%d commands on the callback loop (%.2f ms each) while hipcc ran; fused kernels adopted after %.1f s: %s" % (
    n_cb, t_cb / max(n_cb, 1) * 1e3, adopted_at or -1, c.jit_note), flush=True)
assert adopted_at is not None
for _ in range(10):
    c.command(x)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(200):
    c.command(x)
torch.cuda.synchronize()
print("fused: %.3f ms per command" % ((time.perf_counter() - t1) / 200 * 1e3))
