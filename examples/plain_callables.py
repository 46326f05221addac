"""Your own torch callables, unchanged, on the fused kernels (no gym needed).

    python examples/plain_callables.py [--samples 8192] [--horizon 32] [--steps 200]

`dynamics` / `running_cost` below are ordinary batched torch functions -- exactly what `pytorch_mppi.MPPI` takes.  Built on a
HIP device the controller traces them into a device functor (pytorch_mppi_amd/trace.py), checks the functor against the
callables on random batches, compiles it once with hipcc (1-2 minutes the first time, cached under pytorch_mppi_amd/_jit/)
and from then on runs one fused launch per command; `ctrl.jit_note` says what happened.  `auto_jit=False` keeps the
callback loop (every timestep a handful of ATen launches), for comparison."""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pytorch_mppi_amd import MPPI

DT, GOAL = 0.1, torch.tensor([1.5, -0.5])


def dynamics(state, action):                  # unicycle: (x, y, heading), controls (speed, turn rate)
    x, y, th = state[:, 0], state[:, 1], state[:, 2]
    v, w = action[:, 0], action[:, 1]
    return torch.stack((x + DT * v * torch.cos(th), y + DT * v * torch.sin(th), th + DT * w), dim=1)


def running_cost(state, action):
    d = state[:, :2] - GOAL.to(state.device)
    return (d ** 2).sum(dim=1) + 0.01 * (action ** 2).sum(dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=8192)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    kw = dict(nx=3, noise_sigma=torch.eye(2), num_samples=args.samples, horizon=args.horizon, lambda_=1.0, device="cuda",
              u_min=torch.tensor([-1.0, -2.0]), u_max=torch.tensor([1.0, 2.0]))
    for auto in (True, False):
        ctrl = MPPI(dynamics, running_cost, auto_jit=auto, **kw)
        state = torch.zeros(3, device="cuda")
        for _ in range(5):
            ctrl.command(state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            action = ctrl.command(state)
            state = dynamics(state.view(1, -1), action.view(1, -1))[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"auto_jit={auto}: {ctrl.jit_note or 'generic path (tracer off)'}\n   {dt * 1e6:.0f} us per control step, "
              f"distance to the goal {float((state[:2].cpu() - GOAL).norm()):.3f}")


if __name__ == "__main__":
    main()
