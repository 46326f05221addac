"""Closed-loop swing-up of the gym Pendulum-v1 dynamics with the fused engine (no gym needed).

    python examples/pendulum_closed_loop.py [--samples 8192] [--horizon 32] [--steps 200] [--rng philox]

Same script with `from pytorch_mppi import MPPI` + `device="cpu"` drives the reference."""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pytorch_mppi_amd import MPPI, models


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=8192)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rng", default="torch", choices=["torch", "torch-native", "philox"])
    args = ap.parse_args()

    model = models.Pendulum()                      # torch callables + device functor of tests/pendulum.py
    ctrl = MPPI(model.dynamics, model.running_cost, nx=2, noise_sigma=torch.tensor(10.0), num_samples=args.samples,
                horizon=args.horizon, lambda_=1.0, device="cuda", u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0),
                rng=args.rng)

    def run(steps):
        state = torch.tensor([[math.pi, 1.0]], device="cuda")     # hanging down, spinning
        total = torch.zeros((), device="cuda")
        for _ in range(steps):
            action = ctrl.command(state[0])                       # device tensor, no host sync inside
            state = model.dynamics(state, action.view(1, 1))      # the "environment": same true dynamics
            total = total + model.running_cost(state, action.view(1, 1))[0]      # stays on the device
        torch.cuda.synchronize()
        return state, float(total)

    run(5)                                         # warm-up: kernel loading, allocator
    ctrl.reset()
    t0 = time.perf_counter()
    state, total = run(args.steps)
    dt = time.perf_counter() - t0
    th = ((float(state[0, 0]) + math.pi) % (2 * math.pi)) - math.pi
    print(f"{args.steps} steps in {dt * 1e3:.1f} ms ({dt / args.steps * 1e6:.0f} us per control step incl. env), "
          f"accumulated cost {total:.1f}, final angle {th:+.3f} rad, angular velocity {float(state[0, 1]):+.3f}")


if __name__ == "__main__":
    main()
