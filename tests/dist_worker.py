"""Worker of tests/test_gpu_sharding.py::test_two_processes_*: launched by torch.distributed.run with
2 ranks that SHARE cuda:0 (gloo; RCCL refuses two ranks on one device).  Each rank runs the real
sharded command() -- K1/K3/K4 on its shard, the record all-gather between PROCESSES, K5 -- and
compares with an unsharded controller fed the same noise."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_mppi_amd as pm  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    K, T, nx, nu = 3000, 10, 6, 4
    g = torch.Generator().manual_seed(0)
    U0 = torch.randn(T, nu, generator=g) * 0.05
    x0 = torch.randn(nx, generator=g).cuda()
    m = pm.models.Integrator(nx, nu)
    ok = True
    for rng in ("torch", "philox"):
        kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=20.0, U_init=U0.clone(), sample_null_action=True,
                  u_max=torch.tensor([1.0] * nu), rng=rng, seed=5)
        sharded = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), shard=(rank, world), **kw)
        full = pm.MPPI(m.dynamics, m.running_cost, nx, torch.eye(nu), **kw)
        for step in range(3):
            if rng == "torch":
                z = torch.randn(K, T, nu, generator=g)       # same generator state on every rank
                sharded.inject_noise(z)
                full.inject_noise(z)
            a_s = sharded.command(x0)
            a_f = full.command(x0)
            ok &= bool(torch.allclose(a_s, a_f, rtol=1e-5, atol=1e-6))
            ok &= bool(torch.allclose(sharded.U, full.U, rtol=1e-5, atol=1e-6))
            lo, hi = sharded._shard.bounds(rank)
            ok &= bool(torch.allclose(sharded.omega, full.omega[lo:hi], rtol=1e-4, atol=1e-7))
        # ranks hold bit-identical U
        mine = sharded.U.cpu().reshape(-1)
        both = torch.empty(world * mine.numel())
        dist.all_gather_into_tensor(both, mine)
        ok &= bool(torch.equal(both[:mine.numel()], both[mine.numel():2 * mine.numel()]))
    # KMPPI and SMPPI under sharding (theta / lifted-control updates go through the same record exchange)
    lin = pm.models.LinearGoal(torch.tensor([[1.0, 0.0], [0.0, -1.0]]), torch.tensor([2.0, 2.0]))
    xs = torch.tensor([-3.0, -2.0]).cuda()
    for cls, extra, zshape in ((pm.KMPPI, dict(num_support_pts=5), (K, 5, 2)),
                               (pm.SMPPI, dict(w_action_seq_cost=2.0, delta_t=0.5, action_max=torch.tensor([1.0, 1.0])), (K, T, 2))):
        kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=5.0, U_init=torch.zeros(T, 2), rng="torch", **extra)
        sharded = cls(lin.dynamics, lin.running_cost, 2, torch.eye(2), shard=(rank, world), **kw)
        full = cls(lin.dynamics, lin.running_cost, 2, torch.eye(2), **kw)
        for step in range(3):
            z = torch.randn(*zshape, generator=g)
            sharded.inject_noise(z)
            full.inject_noise(z)
            a_s, a_f = sharded.command(xs), full.command(xs)
            ok &= bool(torch.allclose(a_s, a_f, rtol=1e-5, atol=1e-6))
            ok &= bool(torch.allclose(sharded.U, full.U, rtol=1e-5, atol=1e-6))
        if not ok:
            print(f"rank {rank}: {cls.__name__} sharded mismatch")
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    print(f"rank {rank}: {'OK' if ok else 'MISMATCH'}")
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == "__main__":
    main()
