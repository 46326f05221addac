"""Online-learned dynamics on the fused kernels (no gym needed): the control loop of the reference's
tests/pendulum_approximate.py -- a residual network is fitted to the transitions seen so far and retrained every few steps
while MPPI plans through it.

    python examples/learned_dynamics.py [--samples 1000] [--horizon 30] [--steps 120] [--retrain-every 30]

The network's parameters are run-time inputs of the traced functor (pytorch_mppi_amd/trace.py): `optimizer.step()` between two
commands only re-uploads 1250 numbers, nothing is recompiled.  The first run compiles the kernels in a background thread
(commands use the callback loop until then; `ctrl.wait_for_jit()` below waits for it so that the timing is of the fused path)."""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

from pytorch_mppi_amd import MPPI

DEV = "cuda"
wrap = lambda a: ((a + math.pi) % (2 * math.pi)) - math.pi


def true_dynamics(state, action):             # the plant: a damped pendulum, unknown to the controller
    th, thdot = state[:, 0:1], state[:, 1:2]
    u = torch.clamp(action, -2.0, 2.0)
    thdot = torch.clamp(thdot + (15.0 * torch.sin(th) + 3.0 * u - 0.1 * thdot) * 0.05, -8.0, 8.0)
    return torch.cat((wrap(th + thdot * 0.05), thdot), dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--retrain-every", type=int, default=30)
    args = ap.parse_args()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 2)).to(DEV)

    def dynamics(state, action):              # what MPPI plans through: state + network residual
        u = torch.clamp(action, -2.0, 2.0)
        nxt = state + net(torch.cat((state, u), dim=1))
        nxt[:, 0] = wrap(nxt[:, 0])
        return nxt

    def running_cost(state, action):
        return wrap(state[:, 0]) ** 2 + 0.1 * state[:, 1] ** 2

    def train(data, epochs=150):
        xu, y = data[:-1], data[1:, :2] - data[:-1, :2]
        y[:, 0] = wrap(y[:, 0])
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        for _ in range(epochs):
            opt.zero_grad()
            ((net(xu) - y) ** 2).mean().backward()
            opt.step()

    # bootstrap: random actions on the plant
    state = torch.tensor([[math.pi, 0.0]], device=DEV)
    rows = []
    for _ in range(100):
        u = (torch.rand(1, 1, device=DEV) - 0.5) * 4.0
        rows.append(torch.cat((state, u), dim=1))
        state = true_dynamics(state, u)
    data = torch.cat(rows)
    train(data)

    ctrl = MPPI(dynamics, running_cost, 2, torch.tensor(1.0), num_samples=args.samples, horizon=args.horizon, lambda_=1.0,
                device=DEV, u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))
    print("at construction:", ctrl.jit_note)
    ctrl.wait_for_jit()
    print("now:            ", ctrl.jit_note)
    t_cmd = 0.0
    for i in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u = ctrl.command(state[0])
        torch.cuda.synchronize()
        t_cmd += time.perf_counter() - t0
        rows.append(torch.cat((state, u.view(1, 1)), dim=1))
        state = true_dynamics(state, u.view(1, 1))
        if (i + 1) % args.retrain_every == 0:
            train(torch.cat(rows))             # the controller follows: same functor, new parameter vector
            print(f"step {i + 1}: retrained on {len(rows)} transitions, |theta| = {abs(float(state[0, 0])):.3f}, still {ctrl.jit_note.split(':')[0]}")
    print(f"{t_cmd / args.steps * 1e3:.3f} ms per command (synchronised), final |theta| = {abs(float(state[0, 0])):.3f}")


if __name__ == "__main__":
    main()
