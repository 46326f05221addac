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
        print(f"{wl} state as {name:14s}: {1e3 * (time.perf_counter() - t0) / n:.4f} ms/command", flush=True)
