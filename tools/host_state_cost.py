"""What a host-resident state costs per command (the only host->device transfer of a closed loop): C3, rng=philox,
state handed over as a device tensor, a CPU tensor, a numpy array."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import pytorch_mppi_amd as pm

dev = torch.device("cuda", 0)
for wl in ("c3", "c2"):
    ctrl, x0, _ = bench.make_controller(pm, wl, dev, "philox", None, bench.WORKLOADS[wl][4])
    states = {"device tensor": x0, "cpu tensor": x0.cpu(), "numpy array": x0.cpu().numpy()}
    for name, s in states.items():
        for _ in range(20):
            ctrl.command(s)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            ctrl.command(s)
        torch.cuda.synchronize()
        t_open = (time.perf_counter() - t0) / n
        # closed loop as a simulator sees it: the action goes back to the host every step (a device sync per step)
        t0 = time.perf_counter()
        for _ in range(n):
            a = ctrl.command(s).cpu()
        t_closed = (time.perf_counter() - t0) / n
        print(f"{wl} state as {name:14s}: {1e3 * t_open:.4f} ms/command back to back, {1e3 * t_closed:.4f} ms/step with the action "
              f"read back each step", flush=True)
