"""CPU, world_size 2, gloo: the N>1 path's host logic -- contiguous shard plan, the single
record all-gather, rank-order combine -- gives every rank the same U as the unsharded oracle.
(The per-shard records are produced here by the oracle's pieces; on the GPU they come from
K3/K4 and the combine runs as kernel K5, tested against the same host restatement in -m gpu.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu
from oracle import mppi_oracle as orc
from pytorch_mppi_amd.dist import ShardPlan, combine_records_host


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, d = gu.load(name)
        dtype = gu.TDT[cfg["dtype"]]
        p = gu.oracle_problem(cfg, d)
        U = gu.t(d, "U_init", dtype)
        state = gu.t(d, "state", dtype)
        z = gu.t(d, "z0", dtype)
        full = orc.command(p, U, state, z, True)          # unsharded answer (the checker)
        plan = ShardPlan(cfg["K"], rank, world)
        lo, hi = plan.bounds(rank)
        # this rank's shard record, relative to its OWN minimum
        c = full["cost_total"][lo:hi]
        b = c.min()
        w = torch.exp(-(1 / p.lambda_) * (c - b))
        rec = torch.cat([b.view(1), w.sum().view(1), torch.einsum("k,ktn->tn", w, full["noise"][lo:hi]).reshape(-1)])
        recs = plan.all_gather(rec)                       # the one collective of a command
        recs2, work = plan.all_gather_start(rec)          # the not-waited-for form the controller uses
        if work is not None:
            work.wait()
        assert torch.equal(recs, recs2)
        U_new, beta, eta = combine_records_host(recs, full["U_shifted"], p.lambda_)
        ok = torch.allclose(U_new, full["U"], rtol=1e-10, atol=1e-12) and float(beta) == float(full["beta"])
        # every rank must hold bit-identical results
        both = [torch.empty_like(U_new) for _ in range(world)]
        dist.all_gather(both, U_new)
        same = all(torch.equal(both[0], x) for x in both)
        out_q.put((rank, bool(ok), bool(same), tuple(recs.shape)))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_and_combine_matches_unsharded():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "linear_diag_f64", q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, same, shape in res:
        assert ok and same, (rank, ok, same)
        assert shape == (2, 2 + 10 * 2)


def _worker_replicated_U(rank, world, port, out_q):
    """under `shard=` the randomly initialised nominal sequence (mppi.py:144-145, :290)
    must be ONE draw for all ranks, whatever the ranks' own generator states are."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pytorch_mppi_amd as pm
        torch.manual_seed(100 + rank)                     # deliberately different per rank
        m = pm.models.Integrator(6, 4)
        c = pm.MPPI(m.dynamics, m.running_cost, 6, torch.eye(4), num_samples=64, horizon=7, device="cpu",
                    shard=(rank, world), rng="torch")
        outs = [c.U.clone()]
        c.reset()
        outs.append(c.U.clone())
        same = []
        for u in outs:
            both = [torch.empty_like(u) for _ in range(world)]
            dist.all_gather(both, u)
            same.append(all(torch.equal(both[0], x) for x in both))
        # the per-shard sample generators of the torch modes must differ between ranks
        g = torch.tensor([c._shard_gen.initial_seed()], dtype=torch.int64)
        gs = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(gs, g)
        distinct = len({int(x) for x in gs}) == world
        out_q.put((rank, same, distinct, not torch.equal(outs[0], outs[1])))
    finally:
        dist.destroy_process_group()


def test_sharded_controllers_share_one_initial_sequence_and_draw_distinct_samples():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_replicated_U, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, distinct, changed in res:
        assert all(same), (rank, same)
        assert distinct and changed


def test_bench_self_spawn_plumbing_without_a_gpu():
    """`python bench.py --gpus 2` outside any launcher must start its own two ranks (torch.distributed.run on
    127.0.0.1) and print ONE line from rank 0 -- checked here without a GPU through MPPI_BENCH_SPAWN_ONLY=1 (the ranks
    rendezvous on gloo and agree on the world size); the real thing is tests/test_gpu_sharding.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MPPI_BENCH_SPAWN_ONLY"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"spawn_check": True, "n_gpus": 2}
