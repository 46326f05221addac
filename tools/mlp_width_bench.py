"""C4-shaped commands (K 65536 x T 64, nx 16, nu 4, fp32, rng=philox) at hidden = 64 / 128 / 256: the split-operand 16-bit
MFMA kernel against the exact fp32 MFMA kernel (MPPI_MLP_EXACT=1) and the per-lane VALU kernel (MPPI_MLP_VALU=1).
    python tools/mlp_width_bench.py > gpurun_out/mlp_width_bench.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_mppi_amd as pm

K, T, nx, nu = 65536, 64, 16, 4
x0 = torch.zeros(nx).cuda()
for H in (64, 128, 256):
    row = []
    for name, env in (("split", {}), ("exact", {"MPPI_MLP_EXACT": "1"}), ("valu", {"MPPI_MLP_VALU": "1"})):
        for k in ("MPPI_MLP_EXACT", "MPPI_MLP_VALU"):
            os.environ[k] = env.get(k, "0")
        m = pm.models.MLPResidual.random(nx, nu, H, seed=2)
        c = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=50.0,
                    rng="philox", seed=3)
        for _ in range(3):
            c.command(x0)
        torch.cuda.synchronize()
        n = 20 if name != "valu" else 5
        t0 = time.perf_counter()
        for _ in range(n):
            c.command(x0)
        torch.cuda.synchronize()
        row.append("%s %.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3))
    print("hidden %3d: " % H + " | ".join(row), flush=True)
