"""Host + device cost of the sharded command path (K3 -> record -> all_gather -> K5) against the
single-GPU path (K3 -> K4), measured on ONE GPU with RCCL at world_size 1: the collective has no
peer, so what is timed is launch overhead of the extra stages, not xGMI latency.
    python tools/shard_overhead.py [workload] [rng]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
import bench
import pytorch_mppi_amd as pm

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
rng = sys.argv[2] if len(sys.argv) > 2 else "philox"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
Kper = bench.WORKLOADS[wl][4]
for name, shard, overlap, native in (("single", None, False, "1"), ("sharded(world=1), engine-owned RCCL comm", (0, 1), False, "1"),
                                     ("sharded(world=1), torch.distributed nccl", (0, 1), False, "0"),
                                     ("sharded + next rows behind the collective (torch.distributed)", (0, 1), True, "0")):
    os.environ["MPPI_NATIVE_RCCL"] = native
    ctrl, x0, _ = bench.make_controller(pm, wl, dev, rng, shard, Kper)
    ctrl.lambda_ = 50.0
    ctrl._force_collective = shard is not None
    ctrl.overlap_collective = overlap
    for _ in range(10):
        ctrl.command(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        ctrl.command(x0)
    torch.cuda.synchronize()
    print(f"{name:64s} {1e3 * (time.perf_counter() - t0) / n:.4f} ms/command", flush=True)
    # host-only cost: how long the python call itself takes (the GPU runs behind)
    t0 = time.perf_counter()
    for _ in range(n):
        ctrl.command(x0)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{'':64s} {1e3 * th / n:.4f} ms host time per command", flush=True)
dist.destroy_process_group()
